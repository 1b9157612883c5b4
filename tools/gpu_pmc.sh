#!/bin/bash
# rocprofv3 counter passes over a short bench run (separate passes: TCC slots, SQ slots; no trace domains next to --pmc).
# usage: tools/gpu_pmc.sh <tag> ["pass ..."]   -> gpurun_out/<tag>/pmc_*/p_results.db, profiles-ready summary via tools/pmc_summary.py
set +e
tag=${1:-r02}; out=$PWD/gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
cd /tmp
run() {  # name, counters...
  name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" -d $out/pmc_$name -o p -- python $OLDPWD/bench.py --steps 8 --warmup 2 --chunks 1 --no-extras --no-cpu-baseline > $out/pmc_$name.log 2>&1
}
passes=${2:-"fetch write sq wait mix lds cyc"}
for p in $passes; do
  case $p in
    fetch) run fetch FETCH_SIZE ;;
    write) run write WRITE_SIZE ;;
    sq) run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE ;;
    wait) run wait SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES ;;
    mix) run mix SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA ;;
    lds) run lds SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS ;;
    cyc) run cyc SQ_WAVES SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA ;;
  esac
done
cd $OLDPWD
dbs=$(find $out -name "p_results.db" | sort)
python tools/pmc_summary.py ${tag}_tmp $dbs > $out/pmc_summary_stdout.txt 2>&1
mv profiles/${tag}_tmp_pmc.txt $out/pmc.txt 2>/dev/null; mv profiles/${tag}_tmp_pmc.json $out/pmc.json 2>/dev/null
find $out -name "*.db" -delete
ls -la $out | head -30
