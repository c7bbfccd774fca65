"""CPU oracle for the DiffusionVID inference hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain PyTorch-CPU / numpy / C restatement
of the reference algorithm (sdroh1027/DiffusionVID, `mega_core/...`) and of the three
un-vendored third-party pieces it calls (detectron2 ROIPooler/ROIAlignV2, detectron2
`build_resnet_fpn_backbone`, torchvision `batched_nms`).  Every function cites the
reference file:line it follows.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may
import it -- always as the checker, never as the thing that is measured or shipped.
The product package (`diffusionvid_amd`) never imports it and fails loudly when the HIP
library is missing.

Pinning status (see tests/golden/README.md):
  * schedule, time embedding, DynamicConv, RCNNHead, RCNNHead_cond, DynamicHead
    (extraction + final branches), box<->noise transforms, BoxList.clip_to_image,
    to_image_list, greedy FPS (getGreedyPerm), sampler partitions: PINNED against the
    reference's own modules imported in the build container
    (tests/golden/make_golden.py).
  * ROIAlignV2 / level assignment, R101+FPN, batched_nms: third-party code that is not
    under /root/reference and is un-pinned by the reference (INSTALL.md:68-69 clones
    detectron2 HEAD).  Restated from the published upstream algorithm (SURVEY.md
    Appendix A); "parity unpinned" for those three pieces in the strict sense of the
    task statement.
"""
