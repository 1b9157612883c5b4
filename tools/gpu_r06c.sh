#!/bin/bash
# counters of k_lq_trip: value phase alone (ablation stop 126) and whole kernel
export TMPDIR=/tmp
R=$PWD; out=$R/gpurun_out/r06c; mkdir -p $out
cd /tmp
run() {  # name, extra perf_quick args, counters...
  name=$1; shift; extra=$1; shift
  timeout 300 rocprofv3 --pmc "$@" -d $out/pmc_$name -o p -- python $R/tools/perf_quick.py --lib $R/variants/libhunter_hip_ablate.so $extra > $out/pmc_$name.log 2>&1
}
for mode in "s126:--stop 126" "full:--steps 5"; do
  m=${mode%%:*}; a=${mode#*:}
  run ${m}_fetch "$a" FETCH_SIZE
  run ${m}_write "$a" WRITE_SIZE
  run ${m}_wait "$a" SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE
  run ${m}_mix "$a" SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
  run ${m}_tcc "$a" TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
done
cd $R
for m in s126 full; do
  python tools/pmc_summary.py r06c_$m $(find $out -name "p_results.db" | grep pmc_${m}_ | sort) > /dev/null 2>&1
  mv profiles/r06c_${m}_pmc.txt $out/ 2>/dev/null; rm -f profiles/r06c_${m}_pmc.json
  echo "== $m"; awk '/^k_lq_trip/{p=1;print;next} /^[a-z_A-Z]/{p=0} p' $out/r06c_${m}_pmc.txt
done
find $out -name "*.db" -delete
tail -3 $out/pmc_s126_tcc.log
