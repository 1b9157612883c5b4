export TMPDIR=/tmp
mkdir -p gpurun_out/r02c
timeout 600 python -m pytest tests/test_gpu_failure_surface.py tests/test_lcm_codec.py tests/test_gpu_parity.py -m gpu -q -k "nan_observation or packers or hoqp_two" 2>&1 | tail -80 > gpurun_out/r02c/pytest.log
cd /tmp
for m in "stochastic cycles 1048576" "stochastic time 1" "host_trap time 1" "host_trap instructions 4096"; do set -- $m; echo "== $m" >> $OLDPWD/gpurun_out/r02c/pcs.err; timeout 200 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $1 --pc-sampling-unit $2 --pc-sampling-interval $3 -d $OLDPWD/gpurun_out/r02c/pcs_$1_$2 -o run -- python $OLDPWD/bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline >> $OLDPWD/gpurun_out/r02c/pcs.err 2>&1; done
cd $OLDPWD; rocprofv3 -L 2>&1 | grep -i -A12 "pc.sampl" | head -60 > gpurun_out/r02c/pcs_list.txt; du -sh gpurun_out/r02c/*; find gpurun_out/r02c -name "*.db" -size +20M -delete
