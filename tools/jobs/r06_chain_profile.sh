cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_chain -- python3 examples/frame_pipeline.py > gpurun_out/prof_chain.log 2>&1
f=$(ls -S gpurun_out/prof_chain/*/*_kernel_stats.csv | head -1)
python3 tools/summarize_prof.py stats $f > gpurun_out/r06_chain_kernel_stats.txt 2>&1; head -30 gpurun_out/r06_chain_kernel_stats.txt | cut -c1-200
