# Counter passes (an SQ set, FETCH_SIZE, WRITE_SIZE, GRBM_GUI_ACTIVE; one run each) of the training step's dominant kernels at
# its shape (tools/prof_train_kernels.py: 64 -> 64 single layer and weight gradient at 16 x 1024 x 1024):
# bash tools/profile_r3_train_pmc.sh [tag]
set -o pipefail
out=gpurun_out/${1:-r3_bn}; mkdir -p $out; export TMPDIR=/tmp
i=0
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $CTRS -d $out/t_$i -o p -- python3 tools/prof_train_kernels.py > $out/t_$i.log 2>&1 || { echo fail $i; tail -3 $out/t_$i.log; exit 3; }
done
for i in 1 2 3 4; do python3 tools/rocpd_summary.py pmc $(find $out/t_$i -name "*.db" | head -1) conv3x3; done > $out/train_kernels_pmc.csv
for i in 1 2 3 4; do python3 tools/rocpd_summary.py pmc $(find $out/t_$i -name "*.db" | head -1) wgrad_reduce; done >> $out/train_kernels_pmc.csv
for i in 1 2 3 4; do rm -rf $out/t_$i; done
sed 's/(anonymous namespace):://g' $out/train_kernels_pmc.csv | cut -c1-60,75-300 | grep -v "^kernel"
