"""Command-line surface of the reference's pre-trained mode (gnomix.py:318-355, 359-370, 404-412):

    python -m gnomix_amd <query_file> <output_basename> <chr_nr> <phase> <path_to_model>

`path_to_model` is a flat `.gnx` model (gnomix_amd.GnxModelData.save) or a reference `model.pkl[.gz]`, which is read
with a restricted unpickler (gnomix_amd.refpickle: neither the reference's `src` package nor xgboost is needed; CRF
models still need sklearn_crfsuite's attributes in the pickle) and converted by gnomix_amd.convert.from_reference_model.
Training mode (7-8 arguments) is out of scope.
Outputs: <output_basename>/query_results.msp, .fb (+ .lai, query_results_bed/, query_file_phased.vcf as configured).
"""
from __future__ import annotations

import os
import sys

import numpy as np

USAGE = ("Usage when using a pre-trained model:\n"
         "   $ python3 gnomix.py <query_file> <output_basename> <chr_nr> <phase> <path_to_model>\n"
         "(training a model from scratch is not part of the MI355X inference path; train with the reference and export)")


PLANES_SUFFIX = ".planes"   # <model>.gnx.planes: the logistic base's prepared planes (DeviceModel.export_prepared), a cache


def read_gnx_and_planes(path):
    """the host half of loading a .gnx: the model file and, when present, the cache of its prepared planes -> (data, planes or None).
    Touches no GPU: the command line runs it on a worker thread while the main thread starts the GPU runtime."""
    from .model import GnxModelData
    data = GnxModelData.load(path)
    prepared = None
    side = path + PLANES_SUFFIX
    if os.environ.get("GNX_PLANES_CACHE", "1") != "0" and data.base_kind == "logistic" and os.path.exists(side):
        try:
            prepared = np.fromfile(side, dtype=np.uint8)
        except OSError as e:
            print("ignoring %s: %s" % (side, e), file=sys.stderr)
    return data, prepared


def _load_gnx_with_planes(path, device, verbose, timings=None, file_job=None):
    """A .gnx and, beside it, the cache of its prepared planes (<model>.gnx.planes).  Starting the GPU runtime is the largest part
    of a cold start (0.13-0.2 s of hipInit) and needs nothing of the model: the file reads run on a worker thread meanwhile
    (`file_job`, started by main() before it first touches the GPU).  A cache that does not belong to the model / library / settings
    is reported, ignored and rewritten.  GNX_PLANES_CACHE=0: neither read nor written."""
    from time import perf_counter as clock
    from . import _lib
    from .gnomix import HipGnomix
    t0 = clock()
    if file_job is None:
        file_job = _Prefetch(lambda: read_gnx_and_planes(path))
    ctx = _lib.default_context(device)
    t1 = clock()
    data, prepared = file_job.result()
    use_cache = os.environ.get("GNX_PLANES_CACHE", "1") != "0" and data.base_kind == "logistic"
    side = path + PLANES_SUFFIX
    t2 = clock()
    model = None
    if prepared is not None:
        try:
            model = HipGnomix(data, ctx=ctx, prepared=prepared)
        except _lib.GnxError as e:
            if e.code != _lib.GNX_ESTALE:
                raise
            print("%s is stale (%s): preparing the planes from the model and rewriting it" % (side, e.msg), file=sys.stderr)
            prepared = None
    if model is None:
        model = HipGnomix(data, ctx=ctx)
        if use_cache:
            try:
                blob = model.dev.export_prepared()
                if blob.size:
                    tmp = "%s.tmp%d" % (side, os.getpid())
                    blob.tofile(tmp)
                    os.replace(tmp, side)      # (atomic: a concurrent start reads the old file or the new one, never half of it)
            except OSError as e:               # a read-only model directory: every start prepares the planes, as before
                if verbose:
                    print("not caching the prepared planes (%s)" % e, file=sys.stderr)
    if timings is not None:
        timings.update({"load_model.gpu_runtime": t1 - t0, "load_model.wait_file": t2 - t1, "load_model.device_model": clock() - t2,
                        "load_model.planes_from_cache": float(prepared is not None)})
    return model


def load_model(path_to_model, device=0, verbose=True, timings=None, file_job=None):
    """gnomix.py:26-35 — .gnx directly, .pkl / .pkl.gz through the converter"""
    from .gnomix import HipGnomix
    if verbose:
        print("Loading model...")
    if path_to_model.endswith(".gnx"):
        return _load_gnx_with_planes(path_to_model, device, verbose, timings, file_job)
    from .convert import from_reference_model
    from .refpickle import load_reference_pickle
    # restricted unpickling: the reference's `src` package and xgboost are NOT needed (and nothing of them is executed)
    try:
        ref_model = load_reference_pickle(path_to_model)
    except Exception as e:   # e.g. estimators pickled by an incompatible scikit-learn: fall back to attribute bags
        if verbose:
            print("rebuilding scikit-learn objects failed (%s): reading their attributes only" % type(e).__name__)
        ref_model = load_reference_pickle(path_to_model, use_sklearn=False)
    return HipGnomix(from_reference_model(ref_model), device=device)


class _Prefetch:
    """`fn()` on a worker thread (the library calls release the GIL); result() re-raises what it raised"""

    def __init__(self, fn):
        import threading
        self._out = self._err = None

        def work():
            try:
                self._out = fn()
            except BaseException as e:   # noqa: BLE001 - handed to the caller
                self._err = e
        self._t = threading.Thread(target=work, daemon=True)
        self._t.start()

    def result(self):
        self._t.join()
        if self._err is not None:
            raise self._err
        return self._out


def run_inference(base_args, model, snp_level=False, bed_file_output=False, verbose=False, timings=None, query=None, devices=None):
    """gnomix.py:37-100 with the HIP model behind the same steps.  The query never becomes an (N, C) host matrix: the library
    parses the text into 2-bit rows (gnx_vcf_read), `column_map` keeps vcf_to_npy's bookkeeping (SNP intersection, REF flips,
    absent SNPs) as one int32 per model SNP, the GPU builds X and runs base + smoother (or Gnofix) on it, and the library
    formats .msp / .fb / the phased VCF.  `timings` (a dict) receives the seconds of each stage; `query` = a _Prefetch of
    vcfio.read_vcf started earlier (the command line parses the query while the GPU runtime starts and the model loads).
    `devices` = GPU ordinals (several: the individuals are cut over them, one context and one host thread each, the query parsed
    once and the outputs written once: gnomix_amd/multi.py) or an existing multi.DeviceGroup; None = the model's own device."""
    from time import perf_counter as clock
    from . import postprocess as pp
    from . import vcfio
    query_file, chm, output_path = base_args["query_file"], base_args["chm"], base_args["output_basename"]
    T = timings if timings is not None else {}
    if verbose:
        print("Loading and processing query file...")
    t0 = clock()
    vcf = query.result() if query is not None else vcfio.read_vcf(query_file, chm=chm, ctx=model.dev.ctx)
    assert vcf is not None, "No SNPs of specified chromosome found in query file."
    T["read_vcf"] = clock() - t0
    t0 = clock()
    src, vcf_idx, fmt_idx = vcfio.column_map(vcf, model.snp_pos, model.snp_ref, verbose=verbose)
    N = 2 * vcf.n_samples
    T["column_map"] = clock() - t0
    if verbose:
        print("Inferring ancestry on query data...")
    t0 = clock()
    out = (model.dev.ctx.pinned_empty((N, model.W, model.A), model.dev.proba_dtype()), model.dev.ctx.pinned_empty((N, model.W), np.int32))
    runner = model.dev
    own_group = None
    if devices is not None and not isinstance(devices, (list, tuple)):
        runner = devices                                                   # a DeviceGroup the caller keeps
    elif devices is not None and len(devices) > 1 and vcf.n_samples > 1:
        from .multi import DeviceGroup
        # no more replicas than shards of whole individuals; made on one thread per device; released below whatever happens
        own_group = DeviceGroup(model.data, devices, first=model.dev, n_ind=vcf.n_samples)
        if len(own_group.models) > 1:
            runner = own_group
        T["replicate_model"] = clock() - t0
        t0 = clock()
    try:
        return _run_and_write(model, base_args, runner, vcf, src, vcf_idx, fmt_idx, N, out, T, t0, snp_level, bed_file_output, verbose)
    finally:
        if own_group is not None:
            own_group.close()


def _run_and_write(model, base_args, runner, vcf, src, vcf_idx, fmt_idx, N, out, T, t0, snp_level, bed_file_output, verbose):
    from time import perf_counter as clock
    from . import postprocess as pp
    from . import vcfio
    chm, output_path = base_args["chm"], base_args["output_basename"]
    if not base_args["phase"]:
        proba, labels = runner.infer_gt2(vcf.gt2, N, src, out=out)
        T["infer"] = clock() - t0
    else:
        assert model.smooth is not None, "Smoother is not trained, returning original haplotypes"
        assert model.smooth.gnofix, "Type of Smoother ({}) does not currently support re-phasing".format(model.smooth)
        G_phased, proba, labels, _ = runner.phase_gt2(vcf.gt2, N, src, out_cols=fmt_idx, out=out)
        T["phase"] = clock() - t0
        if verbose:
            print("Writing phased SNPs to disk...")
        t0 = clock()
        rows = np.asarray(vcf_idx) if vcf.rows is None else np.asarray(vcf.rows)[vcf_idx]
        vcfio.write_phased_vcf(vcf, rows, G_phased, output_path + "/" + "query_file_phased", ref=np.asarray(model.snp_ref)[fmt_idx],
                               alt=np.asarray(model.snp_alt)[fmt_idx], headers=vcf.meta_header)
        T["write_vcf"] = clock() - t0
    if verbose:
        print("Saving results...")
    t0 = clock()
    gm_pos, gm_cm = model.data.gen_map_pos, model.data.gen_map_cm
    meta = pp.get_meta_data(chm, model.snp_pos, vcf["variants/POS"], model.W, model.M, gm_pos, gm_cm)
    out_prefix = output_path + "/" + "query_results"
    samples = vcf["samples"]
    pp.write_msp(out_prefix, meta, labels, model.population_order, samples)
    T["write_msp"] = clock() - t0
    t0 = clock()
    pp.write_fb(out_prefix, meta, proba, model.population_order, samples, ctx=model.dev.ctx if os.environ.get("GNX_FB_DEV") == "1" else None)
    T["write_fb"] = clock() - t0
    if snp_level:
        pp.msp_to_lai(out_prefix + ".msp", vcf["variants/POS"], out_prefix + ".lai")
    if bed_file_output:
        pp.msp_to_bed(out_prefix + ".msp", output_path + "/" + "query_results_bed", pop_order=model.population_order)
    return out_prefix


def _since_process_start():
    """seconds since this process was created (Linux: /proc/self/stat field 22 against /proc/uptime)"""
    try:
        with open("/proc/self/stat") as f:
            start_ticks = float(f.read().rsplit(")", 1)[1].split()[19])
        with open("/proc/uptime") as f:
            up = float(f.read().split()[0])
        return up - start_ticks / os.sysconf("SC_CLK_TCK")
    except Exception:
        return float("nan")


def main(argv=None):
    argv = list(sys.argv if argv is None else argv)
    t_startup = _since_process_start() if os.environ.get("GNX_CLI_TIMING") else None
    if "torch" not in sys.modules:
        os.environ.setdefault("GNX_NO_TORCH", "1")   # the command line needs no torch: do not pay for its import
    if len(argv) in (8, 9):
        print("Training mode is not part of this build.\n" + USAGE)
        return 2
    if len(argv) != 6:
        if len(argv) > 1:
            print("Error: Incorrect number of arguments.")
        print(USAGE)
        return 0
    base_args = {"mode": "pre-trained", "query_file": argv[1] if argv[1].strip() != "None" else None,
                 "output_basename": argv[2], "chm": argv[3], "phase": argv[4].lower() == "true", "path_to_model": argv[5]}
    os.makedirs(base_args["output_basename"], exist_ok=True)
    config = {"model": {}, "inference": {}}
    if os.path.exists("./config.yaml"):
        import yaml
        with open("./config.yaml") as f:
            config = yaml.safe_load(f) or config
    print("Launching in pre-trained mode...")
    from time import perf_counter as clock
    query = None
    if base_args["query_file"] and os.environ.get("GNX_CLI_PREFETCH", "1") != "0":
        # neither needs the other: the host's threads parse the text (into pageable memory, no GPU context yet) while this
        # thread starts the GPU runtime (~0.16 s), reads the model and builds its device tables (~0.15 s)
        from . import vcfio
        query = _Prefetch(lambda: vcfio.read_vcf(base_args["query_file"], chm=base_args["chm"], ctx=None))
    t_load = clock()
    file_job, T_load = None, {}
    if base_args["path_to_model"].endswith(".gnx"):   # the model file is read beside the GPU runtime's start (visible_devices is its first use)
        mp = base_args["path_to_model"]
        file_job = _Prefetch(lambda: read_gnx_and_planes(mp))
    from .multi import visible_devices
    devices = visible_devices()        # every GPU of the node (GNX_DEVICES="0,1,.." narrows it): individuals are cut over them
    model = load_model(base_args["path_to_model"], device=devices[0], timings=T_load, file_job=file_job)
    t_load = clock() - t_load
    model.n_cores = (config.get("model") or {}).get("n_cores")            # gnomix.py:365-367
    model.calibrate = (config.get("model") or {}).get("calibrate")
    model.smooth.calibrate = model.calibrate
    model.base.vectorize = True                                           # gnomix.py:370
    if base_args["query_file"]:
        print("Launching inference...")
        inf = config.get("inference") or {}
        T = {"load_model": t_load}
        if t_startup is not None:
            T = {"interpreter_and_imports": t_startup, "load_model": t_load}
        T.update(T_load)
        run_inference(base_args, model, snp_level=bool(inf.get("snp_level_inference")),
                      bed_file_output=bool(inf.get("bed_file_output")), verbose=True, timings=T, query=query, devices=devices)
        if os.environ.get("GNX_CLI_TIMING"):
            T["since_process_start"] = _since_process_start()
            sys.stderr.write("gnomix_amd timings (s): " + ", ".join("%s %.3f" % kv for kv in T.items()) + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
