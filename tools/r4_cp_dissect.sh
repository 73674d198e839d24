mkdir -p gpurun_out/r4
export PM_BENCH_B=16 PM_CASES=celeb
P=/root/repo/posterior_matching_amd/lib/ab
{ echo "== default"; python tools/bench_layer.py
  echo "== direct form (PM_NO_PATCH_CP=1)"; PM_NO_PATCH_CP=1 python tools/bench_layer.py
  for v in 1 2 3 4; do echo "== PM_CP_EXP=$v (1: no MFMA, 2: no weight loads after the first, 3: one patch pass staged only, 4: no epilogue)"; python tools/bench_layer.py $P/libpmhip_cpexp$v.so; done
} > gpurun_out/r4/cp_dissect.txt 2>&1
cat gpurun_out/r4/cp_dissect.txt
