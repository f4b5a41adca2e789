# runtime knobs of the HIP / ROCr stack against the step's 481 + 157 dependent launches (round 6): same box, interleaved with the default
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
rm -f $O/k_*.log $O/k_lines.txt
line() { grep '"metric"' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f value %.0f' % (d['ms_per_step'], d['p50_ms_per_step'], d['value']))"; }
run() { name=$1; shift; ( env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extras $BARGS ) > $O/k_$name.log 2>&1; echo "$name: $(line $O/k_$name.log)" | tee -a $O/k_lines.txt; }
BARGS=""
run default_1 X=1
run dev_kernarg_1 HIP_FORCE_DEV_KERNARG=1
run graph_packet_capture_1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run default_2 X=1
run dev_kernarg_2 HIP_FORCE_DEV_KERNARG=1
run graph_packet_capture_0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run hw_queues_8 GPU_MAX_HW_QUEUES=8
run no_interrupt HSA_ENABLE_INTERRUPT=0
BARGS="--serial"
run serial_default X=1
run serial_dev_kernarg HIP_FORCE_DEV_KERNARG=1
BARGS="--workload lm --batch 1"
run c3_default X=1
run c3_dev_kernarg HIP_FORCE_DEV_KERNARG=1
BARGS="--workload mimi --batch 8"
run c2_default X=1
run c2_dev_kernarg HIP_FORCE_DEV_KERNARG=1
