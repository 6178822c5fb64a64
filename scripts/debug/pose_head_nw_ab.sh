for nw in 4 8 16; do echo "== NW=$nw"; ISLAM_POSE_NW=$nw SKIP_TORCH=1 python scripts/pose_head_bench.py 2>&1 | grep "hip head"; done
