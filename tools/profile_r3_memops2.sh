# Counter passes (FETCH_SIZE, WRITE_SIZE, an SQ set; one run each) of the register-resident LayerNorm kernels and the
# bf16 NHWC warp: bash tools/profile_r3_memops2.sh [tag]
set -o pipefail
out=gpurun_out/${1:-r3_bd}; mkdir -p $out; export TMPDIR=/tmp
for c in ln warp16; do
  i=0
  for CTRS in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    i=$((i+1))
    timeout -s KILL 120 rocprofv3 --pmc $CTRS -d $out/${c}_$i -o p -- python3 tools/prof_new_kernels.py $c > $out/${c}_$i.log 2>&1 || { echo fail $c $i; tail -3 $out/${c}_$i.log; exit 3; }
  done
  pat=layernorm; [ $c = warp16 ] && pat=warp_nhwc
  for i in 1 2 3; do python3 tools/rocpd_summary.py pmc $(find $out/${c}_$i -name "*.db" | head -1) $pat; done > $out/${c}_pmc.csv
  for i in 1 2 3; do rm -rf $out/${c}_$i; done
  cut -c1-50,90-400 $out/${c}_pmc.csv | grep -v "^kernel"
done
