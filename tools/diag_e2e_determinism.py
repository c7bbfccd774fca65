"""Run-to-run determinism of the whole detector on one 120-frame video at INPUT.LOOKAHEAD_BATCHES 15 (the schedule of
tests/test_gpu_e2e.py::test_lookahead_invariance_full_size): K fresh models, detections compared with the first run's."""
import os
import sys

import torch

sys.path.insert(0, ".")


if os.environ.get("DVID_POISON_TORCH"):          # torch.empty() tensors start as NaN / max int: reads of memory nothing wrote show up
    torch.use_deterministic_algorithms(True, warn_only=True)
    torch.utils.deterministic.fill_uninitialized_memory = True


def run(la, frames=120):
    from diffusionvid_amd.config import get_cfg
    from diffusionvid_amd.data.synthetic_video import SyntheticVIDDataset
    from diffusionvid_amd.modeling.detector import build_detection_model
    from diffusionvid_amd.utils import synthetic
    cfg = get_cfg("configs/vid_R_101_DiffusionVID.yaml", ["DTYPE", "float16", "INPUT.LOOKAHEAD_BATCHES", la, "MODEL.DiffusionDet.SAMPLE_STEP", 1], "configs/BASE_RCNN_1gpu.yaml")
    cfg.freeze()
    model = build_detection_model(cfg)
    model.load_state_dict(synthetic.tame_box_deltas(model.state_dict(), 0.1))
    model = model.to("cuda").eval()
    model.noise_fn = synthetic.noise_fn
    ds = SyntheticVIDDataset([frames], cfg, height=600, width=1000, device="cuda", smooth=True)
    res = []
    with torch.no_grad():
        for idx in range(len(ds)):
            res += model(ds[idx][0])
    out = [r.to(torch.device("cpu")) for r in res]
    del model, ds
    torch.cuda.empty_cache()
    return out


def main():
    la = int(sys.argv[1]) if len(sys.argv) > 1 else 15
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    base = run(la)
    for r in range(1, reps):
        got = run(la)
        bad = []
        for f, (a, b) in enumerate(zip(base, got)):
            if len(a) != len(b) or not torch.equal(a.get_field("labels"), b.get_field("labels")) or not torch.equal(a.bbox, b.bbox) \
                    or not torch.equal(a.get_field("scores"), b.get_field("scores")):
                bad.append(f)
        print("look-ahead %d run %d: %d of %d frames differ %s" % (la, r, len(bad), len(base), bad[:20]), flush=True)




def same(a, b):
    return [f for f, (x, y) in enumerate(zip(a, b)) if len(x) != len(y) or not torch.equal(x.get_field("labels"), y.get_field("labels"))
            or not torch.allclose(x.bbox, y.bbox, atol=1e-3, rtol=0)]


def sequence():
    """the test's own order in one process: look-ahead 1, 6, 15, three times over"""
    first = {}
    for rep in range(3):
        outs = {la: run(la) for la in (1, 6, 13)}
        for la in (6, 13):
            print("rep %d: look-ahead %d vs 1: frames that differ %s" % (rep, la, same(outs[1], outs[la])[:20]), flush=True)
        for la in (1, 6, 13):
            if la in first:
                print("rep %d: look-ahead %d vs its first run: frames that differ %s" % (rep, la, same(first[la], outs[la])[:20]), flush=True)
            else:
                first[la] = outs[la]




def per_config():
    """look-ahead 15 with every igemm2 tile configuration forced in turn (wherever it is valid) against the tuner's own choice"""
    from diffusionvid_amd import _lib
    lib = _lib.load()
    _lib.check(lib.dvid_igemm_set_config(0), "set_config")          # (a forced configuration also means: every layer on igemm2)
    base = run(15)
    for cfg in range(1, lib.dvid_igemm_num_configs()):
        _lib.check(lib.dvid_igemm_set_config(cfg), "set_config")
        for rep in range(2):
            got = run(15)
            print("configuration %d forced (run %d): frames that differ from configuration 0's run %s" % (cfg, rep, same(base, got)[:20]), flush=True)
    lib.dvid_igemm_set_config(-1)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "save":          # detections of one look-ahead-13 run to a file (compare across environments)
        out = run(int(sys.argv[3]) if len(sys.argv) > 3 else 13)
        torch.save([(o.bbox, o.get_field("scores"), o.get_field("labels")) for o in out], sys.argv[2])
    elif len(sys.argv) > 1 and sys.argv[1] == "cmp":
        a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
        bad = [f for f, (x, y) in enumerate(zip(a, b)) if any(p.shape != q.shape or not torch.equal(p, q) for p, q in zip(x, y))]
        print("%s vs %s: %d of %d frames differ %s" % (sys.argv[2], sys.argv[3], len(bad), len(a), bad[:20]))
    elif len(sys.argv) > 1 and sys.argv[1] == "seq":
        sequence()
    elif len(sys.argv) > 1 and sys.argv[1] == "cfg":
        per_config()
    else:
        main()
