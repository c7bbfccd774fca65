for pf in 0 1; do
DVID_IGEMM_PREFETCH_A=$pf python tools/bench_igemm.py --iters 10 --batch 24 2>/dev/null | grep -E "conv3|shortcut|backbone total|res3.0.conv1|res4.0.conv1|dynamic_layer|in_proj|linear1" > gpurun_out/layers_pf$pf.txt
done
