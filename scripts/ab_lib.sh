for i in 1 2 3; do
for lib in prev hip; do
ISLAM_HIP_LIB=$PWD/islam_amd/lib/libislam_$lib.so timeout 300 python bench.py --no-frontend --no-cpu-baseline > gpurun_out/ab.json 2> gpurun_out/ab.err; python -c "
import json;d=json.load(open('gpurun_out/ab.json'));print('$lib', round(d['us_per_lm_iter'],2), d['roofline']['solve_launch_us']['root_L4+downsweep'])"
done; done
