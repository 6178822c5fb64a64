for rep in 1 2; do
  for lib in islam_amd/lib/libislam_hip.so islam_amd/lib/libislam_probe_old.so; do
    ISLAM_HIP_LIB=$lib python3 scripts/vio_only.py 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
g=d['diagnostics']['gpu_side_ms_per_step']
print('$lib  pipelined %.1f f/s (%.3f ms)  sequential %.1f  forward-only %.1f  replay %.3f / %.3f ms (pipelined / sequential)' % (d['value'], d['ms_per_batch'], d['sequential_frames_per_s'], d['forward_only_frames_per_s'], g['pipelined']['frozen_replay_gpu_ms'], g['sequential']['frozen_replay_gpu_ms']))"
  done
done
