export TMPDIR=/tmp
python tools/perf_quick.py --steps 20
for v in hunter_bipedal_control_amd/libvariant_*.so; do timeout 120 python tools/perf_quick.py --steps 20 --lib $v; done
