set -e
mkdir -p gpurun_out/r2p
O=gpurun_out/r2p
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --dtype bf16 --no-cpu-baseline > $O/bench_n1_bf16.json 2>/dev/null
python bench.py --batch 2 --no-roofline --no-cpu-baseline > $O/bench_b2.json 2>/dev/null
python bench.py --config 1 --no-cpu-baseline > $O/bench_config1.json 2>/dev/null
python bench.py --config 1 --graph --no-cpu-baseline > $O/bench_config1_graph.json 2>/dev/null
python bench.py --config 3 --no-cpu-baseline > $O/bench_config3.json 2>/dev/null
python bench.py --config 4 --no-cpu-baseline > $O/bench_config4.json 2>/dev/null
python bench.py --config 5 --no-cpu-baseline > $O/bench_config5.json 2>/dev/null
echo benches done
bash tools/collect_profiles.sh $O round2
echo fp32 profile done
bash tools/collect_profiles.sh $O round2_bf16 --dtype bf16
echo bf16 profile done
ROOT=$(pwd)
( cd /tmp && export TMPDIR=/tmp && RUNET_NO_WGRAD_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/ss_trace -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1 )
python3 tools/prof_summary.py $O/ss_trace 13 > $O/round2_kernel_stats_single_stream.txt
cp $(ls $O/ss_trace/*/*kernel_stats.csv | head -1) $O/round2_kernel_stats_single_stream.csv
rm -rf $O/ss_trace
echo ss done
python tools/conv_launches.py > $O/round2_conv_launches.txt 2>/dev/null
python tools/conv_launches.py --dtype bf16 > $O/round2_bf16_conv_launches.txt 2>/dev/null
echo all done
