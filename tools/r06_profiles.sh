ulimit -c 0
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r06_split_b4 --streams 1 > /dev/null 2>&1
# one frame per forward, one stream: kernel trace only
O=$GRAFT_REPO_ROOT/gpurun_out/prof_r06_split_b1; mkdir -p $O
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 48 --warmup 4 --repeats 1 --no-cpu-baseline --no-latency-mode --no-parity-mode --no-other-configs --no-cpp-host --streams 1 --batch 1 > $O/trace.log 2>&1)
python tools/prof_summary.py $O/trace/bench_results.db > $O/kernel_trace.txt 2>&1; rm -rf $O/trace
bash tools/pmc_kernel.sh r06frame conv_rows_kernelILi8,conv_rows_kernelILi4,encoder_mlp_stream_kernel,linear_split_resident_kernel,set_attention_split_kernel,pfn_kernel,conv_f16_kernel,conv1x1_resident_split -- python $GRAFT_REPO_ROOT/bench.py --steps 24 --warmup 3 --repeats 1 --no-cpu-baseline --no-latency-mode --no-parity-mode --no-other-configs --no-cpp-host --no-graph --streams 1 > /dev/null 2>&1
python tools/pmc_counters_summary.py gpurun_out/pmc_r06frame conv_rows_kernelILi8,conv_rows_kernelILi4,encoder_mlp_stream_kernel,linear_split_resident_kernel,set_attention_split_kernel,pfn_kernel,conv_f16_kernel,conv1x1_resident_split > gpurun_out/r06_frame_counters.txt 2>&1
python tools/conv_layers.py 4 split > gpurun_out/r06_conv_layers.txt 2>&1
python tools/conv_layers.py 1 split >> gpurun_out/r06_conv_layers.txt 2>&1
python bench.py > gpurun_out/r06_default_bench_line.json 2> gpurun_out/r06_default_bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_driver_bench_line.json 2> gpurun_out/r06_driver_bench.err
tail -3 gpurun_out/r06_frame_counters.txt; tail -2 gpurun_out/r06_conv_layers.txt
