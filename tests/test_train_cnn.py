"""SURVEY.md §8 f4 — training the convolutional smoother (CNN.fit, reference src/Smooth/cnn.py:104-118).

  CPU   the oracle's float32 restatement (oracle.cnn_fit) vs golden G17 = the reference's OWN CNN.fit run under torch on 300 rows
        (batches of 128 / 128 / 44, the DataLoader's recorded row order, 12 epochs): trained parameters within 1e-6;
  GPU   gnx_train_cnn vs G17 and vs the oracle (a second geometry with the default shape A = 7, S = 75 of the "large" mode), through
        HipSmoother.train end to end, plus the argument checks of the C ABI.
Adam divides by sqrt(v): a parameter moves by ~lr per step whatever the size of its gradient, so float32 summation-order
differences between backends stay at the 1e-7 level instead of being amplified (measured: oracle vs torch 3e-8).
"""
import numpy as np
import pytest

from conftest import load_golden


def test_oracle_cnn_fit_vs_reference_G17(oracle):
    g = load_golden("G17_cnn_train.npz")
    w, b, loss = oracle.cnn_fit(g["B"], g["y"], g["w0"], g["b0"], int(g["epochs"]), batch=128, order=g["order"])
    assert np.abs(g["w1"] - g["w0"]).max() > 0.02                       # the fit moved the parameters
    assert np.abs(w - g["w1"]).max() < 1e-6 and np.abs(b - g["b1"]).max() < 1e-6
    assert np.all(np.diff(loss) < 0)                                     # the loss the reference prints falls every epoch here
    p, _ = oracle.smooth_cnn(g["B"][:16], w, b)
    assert np.abs(p - g["proba16"]).max() < 1e-6


def test_cnn_init_matches_torch_bounds():
    from gnomix_amd.train import cnn_init
    w, b = cnn_init(7, 75, seed=3)
    bound = 1.0 / np.sqrt(7 * 75)
    assert w.shape == (7, 7, 75) and b.shape == (7,) and w.dtype == np.float32
    assert np.abs(w).max() <= bound and np.abs(b).max() <= bound and np.abs(w).max() > 0.9 * bound


@pytest.mark.gpu
def test_hip_cnn_fit_vs_reference_G17(oracle):
    from gnomix_amd.train import train_cnn_arrays
    g = load_golden("G17_cnn_train.npz")
    w, b, loss = train_cnn_arrays(g["B"], g["y"], int(g["S"]), weight=g["w0"], bias=g["b0"], max_ep=int(g["epochs"]), order=g["order"])
    assert np.abs(w - g["w1"]).max() < 2e-6 and np.abs(b - g["b1"]).max() < 2e-6
    _, _, loss_o = oracle.cnn_fit(g["B"], g["y"], g["w0"], g["b0"], int(g["epochs"]), batch=128, order=g["order"])
    assert np.allclose(loss, loss_o, rtol=0, atol=2e-6)
    # float64 base probabilities (what Gnomix.train hands over) are converted like torch.tensor(B, dtype=torch.float)
    w64, b64, _ = train_cnn_arrays(g["B"].astype(np.float64), g["y"], int(g["S"]), weight=g["w0"], bias=g["b0"], max_ep=int(g["epochs"]),
                                   order=g["order"])
    assert np.array_equal(w64, w) and np.array_equal(b64, b)


@pytest.mark.gpu
@pytest.mark.parametrize("A,S,W,N,batch,epochs", [(7, 75, 130, 200, 128, 4), (3, 5, 17, 9, 4, 6), (12, 31, 64, 70, 32, 3)])
def test_hip_cnn_fit_vs_oracle(oracle, A, S, W, N, batch, epochs):
    from gnomix_amd.train import cnn_init, train_cnn_arrays
    rng = np.random.RandomState(A * 100 + S)
    y = rng.randint(A, size=(N, W)).astype(np.int32)
    B = rng.dirichlet(np.ones(A) * 0.5, size=(N, W)).astype(np.float32)
    w0, b0 = cnn_init(A, S, seed=5)
    order = np.stack([rng.permutation(N) for _ in range(epochs)])
    w, b, loss = train_cnn_arrays(B, y, S, weight=w0, bias=b0, max_ep=epochs, batch_size=batch, order=order)
    wo, bo, lo = oracle.cnn_fit(B, y, w0, b0, epochs, batch=batch, order=order)
    assert np.abs(w - wo).max() < 5e-6 and np.abs(b - bo).max() < 5e-6
    assert np.allclose(loss, lo, rtol=0, atol=5e-6)
    # no `order`: rows in file order every epoch
    w2, b2, _ = train_cnn_arrays(B, y, S, weight=w0, bias=b0, max_ep=2, batch_size=batch, shuffle=False)
    wo2, bo2, _ = oracle.cnn_fit(B, y, w0, b0, 2, batch=batch)
    assert np.abs(w2 - wo2).max() < 5e-6 and np.abs(b2 - bo2).max() < 5e-6


@pytest.mark.gpu
def test_hip_smoother_train_cnn_end_to_end(oracle):
    """HipSmoother.train on a "cnn" model (Smoother.train -> CNN.fit): the trained layer serves predict_proba at once and
    reduces the training loss; predictions equal the oracle's CNN run on the trained parameters"""
    import gnomix_amd as ga
    from gnomix_amd.train import cnn_init
    g = load_golden("G17_cnn_train.npz")
    A, S, W = int(g["A"]), int(g["S"]), int(g["W"])
    d = ga.GnxModelData(C=W * 10 + 3, M=10, A=A, S=S, context=0, smooth_kind="cnn")
    d.cnn_weight, d.cnn_bias = cnn_init(A, S, seed=1)
    sm = ga.HipSmoother(ga.DeviceModel(d))
    acc0 = np.mean(sm.predict(g["B"]) == g["y"])
    sm.train(g["B"], g["y"], max_ep=40, seed=2)
    assert sm.train_loss[-1] < sm.train_loss[0] - 0.1
    acc1 = np.mean(sm.predict(g["B"]) == g["y"])
    assert acc1 > acc0 + 0.1 and acc1 > 0.6
    p_o, l_o = oracle.smooth_cnn(g["B"], sm.dev.data.cnn_weight, sm.dev.data.cnn_bias)
    assert np.abs(sm.predict_proba(g["B"]) - p_o).max() < 1e-5


@pytest.mark.gpu
def test_train_cnn_rejects_bad_arguments():
    import gnomix_amd as ga
    from gnomix_amd.train import train_cnn_arrays
    B = np.full((4, 10, 3), 1 / 3, np.float32)
    y = np.zeros((4, 10), np.int32)
    with pytest.raises(ga.GnxError, match="odd kernel size"):
        train_cnn_arrays(B, y, 4, weight=np.zeros((3, 3, 4), np.float32), bias=np.zeros(3, np.float32), max_ep=1)
    y[0, 0] = 3
    with pytest.raises(ga.GnxError, match="label outside"):
        train_cnn_arrays(B, y, 5, max_ep=1, seed=0)
    y[0, 0] = 0
    with pytest.raises(ga.GnxError, match="row index outside"):
        train_cnn_arrays(B, y, 5, max_ep=1, seed=0, order=np.full((1, 4), 4))
    with pytest.raises(ValueError, match="order must be"):
        train_cnn_arrays(B, y, 5, max_ep=2, seed=0, order=np.zeros((1, 4), np.int64))
