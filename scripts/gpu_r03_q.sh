# round 3, call q: the whole GPU suite + smoke on the current tree
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/q_smoke.log 2>&1; echo "smoke rc=$?"
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 > $O/q_pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -15 $O/q_pytest_gpu.log
