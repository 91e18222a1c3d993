"""-m gpu: the logistic base's prepared planes as a blob (gnx_model_export_prepared -> gnx_model_desc.prepared) and the command
line's cache of them beside the model file (<model>.gnx.planes; VERDICT r5 item 5: a cold start prepared both plane sets on every
run).  Bar: a model loaded from its blob gives BIT-identical outputs; a blob of another model, another ABI, other plane settings, a
truncated or a garbage blob is refused with GNX_ESTALE and a message (nothing half-loaded); the command line reports a stale cache,
ignores it and rewrites it — its outputs stay byte-identical."""
import os
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ga():
    import gnomix_amd
    gnomix_amd.load_library()
    return gnomix_amd


@pytest.mark.parametrize("C,M,A,smooth", [(6037, 100, 7, "xgb"), (3001, 100, 12, "crf"), (2531, 100, 3, "cnn")])
def test_model_from_its_prepared_planes_is_bit_identical(ga, monkeypatch, C, M, A, smooth):
    from gnomix_amd import synth, _lib
    monkeypatch.setenv("GNX_LR_P2", "2")          # both plane sets (small geometries would skip the 2-bit ones)
    d = synth.synthetic_model(C=C, M=M, A=A, S=11, n_rounds=4, seed=C, smooth=smooth)
    X = synth.synthetic_X(37, C, seed=2, miss=0.03)
    ctx = _lib.Context(0)
    m0 = ga.DeviceModel(d, ctx=ctx)
    blob = m0.export_prepared()
    assert blob.dtype == np.uint8 and blob.size > 128 and bytes(blob[:7]) == b"GNXPLR1"
    m1 = ga.DeviceModel(d, ctx=ctx, prepared=blob)
    assert np.array_equal(m1.export_prepared(), blob)              # what was loaded is what was exported
    for a, b in ((m0.infer(X), m1.infer(X)), (m0.infer_packed(m0.pack_x(X)), m1.infer_packed(m1.pack_x(X)))):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    b0, b1 = m0.base_predict(X), m1.base_predict(X)
    assert all((x is None and y is None) or np.array_equal(x, y) for x, y in zip(b0, b1))

    def refused(bad, what):
        with pytest.raises(_lib.GnxError) as e:
            ga.DeviceModel(d, ctx=ctx, prepared=bad)
        assert e.value.code == _lib.GNX_ESTALE and "prepared planes" in e.value.msg and what in e.value.msg, e.value.msg

    refused(blob[: blob.size - 1], "truncated")
    refused(blob[:64], "truncated")
    refused(np.concatenate([blob, np.zeros(16, np.uint8)]), "size")
    g = blob.copy(); g[:8] = np.frombuffer(b"NOTGNXPL", np.uint8)
    refused(g, "not a prepared-planes blob")
    g = blob.copy(); g[12:16] = np.frombuffer(struct.pack("<I", _lib.GNX_ABI_VERSION + 1), np.uint8)
    refused(g, "ABI")
    d2 = synth.synthetic_model(C=C, M=M, A=A, S=11, n_rounds=4, seed=C + 1, smooth=smooth)   # the same geometry, other weights
    with pytest.raises(_lib.GnxError) as e:
        ga.DeviceModel(d2, ctx=ctx, prepared=blob)
    assert e.value.code == _lib.GNX_ESTALE and "other coefficients" in e.value.msg
    d3 = synth.synthetic_model(C=C + 100, M=M, A=A, S=11, n_rounds=4, seed=C, smooth=smooth)
    with pytest.raises(_lib.GnxError) as e:
        ga.DeviceModel(d3, ctx=ctx, prepared=blob)
    assert e.value.code == _lib.GNX_ESTALE and "geometry" in e.value.msg
    # other plane settings: the same model loaded without the 2-bit planes
    monkeypatch.setenv("GNX_LR_P2", "0")
    with pytest.raises(_lib.GnxError) as e:
        ga.DeviceModel(d, ctx=ctx, prepared=blob)
    assert e.value.code == _lib.GNX_ESTALE and "GNX_LR_P2" in e.value.msg
    assert ga.DeviceModel(d, ctx=ctx).export_prepared().size < blob.size
    # nothing was half-loaded by the refusals: the context still works
    assert np.array_equal(m0.infer(X)[1], m1.infer(X)[1])
    # a model without a logistic base has nothing to prepare
    dn = ga.GnxModelData(C=2003, M=10, A=3, S=5, context=5, smooth_kind="xgb")
    for k, v in synth.synthetic_trees(2, 3, 15, seed=1).items():
        setattr(dn, k, v)
    assert ga.DeviceModel(dn, ctx=ctx).export_prepared().size == 0
    ctx.close()


def test_command_line_keeps_and_checks_the_planes_cache(ga, tmp_path, capfd):
    from gnomix_amd import synth, cli, vcfio
    d = synth.synthetic_model(C=8037, M=100, A=5, S=21, n_rounds=6, seed=5)
    d.snp_pos = 1000 + 37 * np.arange(d.C)
    d.snp_ref = np.array(["A"] * d.C)
    d.snp_alt = np.array(["C"] * d.C)
    d.gen_map_pos = np.array([1, 400_000])
    d.gen_map_cm = np.array([0.0, 1.3])
    mp = str(tmp_path / "model.gnx")
    d.save(mp)
    X = synth.synthetic_X(8, d.C, seed=1, miss=0.02)
    vcf = synth.write_vcf_gt2(str(tmp_path / "q.vcf"), vcfio.pack_gt2(X), 4, d.snp_pos, d.snp_ref, d.snp_alt, chrom="22")
    side = mp + cli.PLANES_SUFFIX

    def run(tag):
        out = str(tmp_path / tag)
        assert cli.main(["gnomix.py", vcf, out, "22", "False", mp]) == 0
        return open(out + "/query_results.msp", "rb").read(), open(out + "/query_results.fb", "rb").read()

    assert not os.path.exists(side)
    first = run("o1")
    assert os.path.exists(side) and open(side, "rb").read(7) == b"GNXPLR1"       # written by the first start
    blob = open(side, "rb").read()
    t0 = os.path.getmtime(side)
    capfd.readouterr()
    assert run("o2") == first and open(side, "rb").read() == blob                  # used, not rewritten
    assert "stale" not in capfd.readouterr().err
    assert os.path.getmtime(side) == t0
    # a truncated cache and the cache of another model: reported, ignored, rewritten; outputs unchanged
    open(side, "wb").write(blob[: len(blob) // 2])
    assert run("o3") == first
    assert "is stale" in capfd.readouterr().err and open(side, "rb").read() == blob
    d2 = synth.synthetic_model(C=8037, M=100, A=5, S=21, n_rounds=6, seed=6)
    other = ga.DeviceModel(d2).export_prepared()
    other.tofile(side)
    assert run("o4") == first
    assert "other coefficients" in capfd.readouterr().err and open(side, "rb").read() == blob
    # GNX_PLANES_CACHE=0: neither read nor written
    os.remove(side)
    os.environ["GNX_PLANES_CACHE"] = "0"
    try:
        assert run("o5") == first and not os.path.exists(side)
    finally:
        del os.environ["GNX_PLANES_CACHE"]
