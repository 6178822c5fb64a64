for k in 1 2 3; do echo "== ISLAM_POSE_SPLIT_BIG=$k"; ISLAM_POSE_SPLIT_BIG=$k SKIP_TORCH=1 python scripts/pose_head_bench.py 2>&1 | grep "graphs=1"; done
