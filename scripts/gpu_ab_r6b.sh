# Round 6, call B: (1) k_gemm_xp_once on the depth transformer's linear_out alone (MMI_GEMM_ONCE=0 = off), (2) the codec streams
# of the duplex pipeline confined to n CUs (MMI_DUPLEX_CODEC_CUS), (3) the pipeline's device-clock timeline.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
line() { grep '"metric"' $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f p50 %.3f frames/s %.0f' % (d['ms_per_step'], d['p50_ms_per_step'], d['value']))"; }
run() {
  name=$1; shift
  ( env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extras $BARGS ) > $O/ab_$name.log 2>&1
  echo "$name: $(line $O/ab_$name.log)" | tee -a $O/ab_lines_b.txt
}
rm -f $O/ab_lines_b.txt
BARGS="--serial"
run serial_once_off MMI_GEMM_ONCE=0
run serial_once_on X=1
run serial_once_off2 MMI_GEMM_ONCE=0
run serial_once_on2 X=1
BARGS=""
run pipe_base MMI_GEMM_ONCE=0
for n in 64 96 128 160 192 224; do run pipe_cus$n MMI_GEMM_ONCE=0 MMI_DUPLEX_CODEC_CUS=$n; done
run pipe_base2 MMI_GEMM_ONCE=0
timeout 300 python scripts/duplex_timeline.py 2>&1 | grep -v amdgpu.ids > $O/duplex_timeline_base.txt
MMI_DUPLEX_CODEC_CUS=128 timeout 300 python scripts/duplex_timeline.py 2>&1 | grep -v amdgpu.ids > $O/duplex_timeline_cus128.txt
tail -12 $O/duplex_timeline_base.txt | cut -c1-220
tail -12 $O/duplex_timeline_cus128.txt | cut -c1-220
