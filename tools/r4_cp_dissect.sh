# Dissection of patch_conv_cp_bf16_kernel (DESIGN.md section 6.R4 "CelebA"): tools/pm_cp_exp_dissection.patch adds -DPM_CP_EXP=n
# variants (1: no MFMA, 2: no weight loads after the first, 3: one patch pass staged only, 4: no epilogue).  Build them HERE first:
#   git apply tools/pm_cp_exp_dissection.patch && for v in 1 2 3 4; do tools/build_variant.sh cpexp$v pm_conv.hip -DPM_CP_EXP=$v; done
#   git apply -R tools/pm_cp_exp_dissection.patch
# then run this script through gpurun (the variant libraries travel with the snapshot).
mkdir -p gpurun_out/r4
export PM_BENCH_B=16 PM_CASES=celeb
P=/root/repo/posterior_matching_amd/lib/ab
{ echo "== default"; python tools/bench_layer.py
  echo "== direct form (PM_NO_PATCH_CP=1)"; PM_NO_PATCH_CP=1 python tools/bench_layer.py
  for v in 1 2 3 4; do echo "== PM_CP_EXP=$v (1: no MFMA, 2: no weight loads after the first, 3: one patch pass staged only, 4: no epilogue)"; python tools/bench_layer.py $P/libpmhip_cpexp$v.so; done
} > gpurun_out/r4/cp_dissect.txt 2>&1
cat gpurun_out/r4/cp_dissect.txt
