// Library-wide options: ONE table, set through the C ABI (dvid_set_option / the dvid_igemm_set_* / dvid_set_* entry points of
// include/dvid_hip.h), never through the environment.  The defaults below are the benchmarked configuration; dvid_effective_config
// prints the table, bench.py echoes it and tests/test_host_logic.py pins the defaults.  (Round 6: these replaced 31 DVID_* environment
// switches, most of them A/B paths that had lost their measurement.  What is still read from the environment: DVID_LIB, DVID_IGEMM_TUNE,
// DVID_IGEMM_TUNE_CACHE, DVID_POISON_WORKSPACE, DVID_CHAINS, DVID_CALL_GRAPH, DVID_PROFILE_DUMP -- see INTEGRATION.md.)
#pragma once

struct DvidOptions {
    // kernel-family choices: 1 = by the layer's shape rule, 2 = wherever the layer type fits (tests), 0 = off (the igemm2 kernel / separate launches)
    int conv3x3 = 1;         // 3x3 / stride-1 layers on the halo-staged patch kernels (csrc/conv3x3.hip)
    int wstat = 1;           // short-K / wide-N 1x1 layers on the weight-stationary kernels (csrc/wstat.hip)
    int bneck_fuse = 1;      // res2 / res3 bottleneck tails as one launch (csrc/bneck.hip)
    int stem_pool = 1;       // stem + ReLU + max pool as one launch (csrc/conv3x3.hip: stem_pool_kernel); 0 / 1
    int head_tail = 1;       // FFN .. apply_deltas of a head pass as one row-tile kernel (csrc/headtail.hip); 0 = layer by layer
    int roi_fuse = 1;        // RoIAlign gathered straight into DynamicConv's LDS tile (csrc/dynconv.hip, one launch instead of two; bit-identical) wherever a head pass gets its proposal features from the caller; 0 / 1
    int ln_rows = 1;         // LayerNorm at d = 128 / 256: several rows per wave (bit-identical to one row per wave); 0 / 1
    int igemm_cfg = -1;      // >= 0: force this igemm2 tile configuration wherever it is valid (bit-identity tests); -1 = the tuner
    int igemm_tune = -1;     // 1 = time new shape buckets, 0 = never time, -1 = DVID_IGEMM_TUNE if set, else 0 when DVID_IGEMM_TUNE_CACHE holds winners, else 1
    int igemm_generic = 0;   // 1 = igemm2's general addressing + general epilogue (the specialised paths are compared with it bit for bit)
    int f32_split = 1;       // DTYPE float32 products: 1 = split (hi, lo) fp16 operands, three fp16-MFMA passes (csrc/f32.hip: f32x3_igemm_kernel); 0 = exact fp32 products on the fp32 MFMA
    int f32_wstat = 1;       // DTYPE float32, split operands: short-K / wide-N 1x1 layers on the weight-stationary kernel (csrc/f32_wstat.hip); 1 = by the shape rule, 2 = wherever it fits, 0 = off
    int f32_conv3x3 = 1;     // DTYPE float32, split operands: 3x3 / stride-1 layers on the halo-staged kernel (csrc/f32_conv3x3.hip); 0 = the tiled kernel
    int bneck_lds = 0;       // diagnostics: dynamic LDS bytes of the fused block kernels (0 = the whole 160 KB, so nothing with LDS shares their CU)
};
extern DvidOptions g_opt;
