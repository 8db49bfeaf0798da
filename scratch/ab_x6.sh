#!/bin/bash
# ablation builds of the bf16x6 forward (timing only): each variant is a full library selected with PTR_LIB
cd /root/repo
for v in NODMA NOBAR NOX NOSTORE; do
  python -m ptranking_amd.build --variant x6_$v PTR_X6_$v --src scorer_x6.hip > /dev/null 2>&1 || echo "build $v failed"
done
python -m ptranking_amd.build --variant x6_NODMA_NOBAR PTR_X6_NODMA PTR_X6_NOBAR --src scorer_x6.hip > /dev/null 2>&1
ls -la ptranking_amd/*.so
