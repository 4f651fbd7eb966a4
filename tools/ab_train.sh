#!/bin/bash
# A/B of the training step over library variants in superresolution_amd/_exp (GPU box).
for v in "$@"; do
  if [ "$v" = "cur" ]; then unset SRHIP_LIBRARY; else export SRHIP_LIBRARY=$PWD/superresolution_amd/_exp/libsrhip_$v.so; fi
  echo -n "$v: "
  timeout -k 10 200 python bench.py --workload train --steps 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), 'ms/step')" || exit 1
done
