mkdir -p gpurun_out/r4
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -k "masked or (layer_fwd and 256)" > gpurun_out/r4/cp_test.log 2>&1; tail -3 gpurun_out/r4/cp_test.log
export PM_BENCH_B=16 PM_CASES=celeb
for v in "" "PM_NO_PATCH_CP=1"; do echo "== $v"; env $v python tools/bench_layer.py 2>/dev/null; done > gpurun_out/r4/cp_layers.txt 2>&1; cat gpurun_out/r4/cp_layers.txt
tools/ab_workload.sh pm_vqvae_celeb_a 16 "-" "PM_NO_PATCH_CP=1" "-" "PM_NO_PATCH_CP=1" > gpurun_out/r4/ab_cp.txt 2>&1; cat gpurun_out/r4/ab_cp.txt
