# usage (GPU box, from the repo root): bash tools/refresh_round_profiles.sh [round tag, default round3]
# Every committed measurement of a round on ONE box, on the code as it stands: bench lines of the five BASELINE configurations (+ bf16 and the
# 2-image shard), rocprofv3 kernel tables (two streams / single stream), FETCH_SIZE / WRITE_SIZE passes per workload (the files bench.py's
# `traffic` / `hbm` fields quote: keyed by model, operand type and batch shape), per-launch tables.  Copy gpurun_out/<tag>p/* to profiles/.
set -e
TAG=${1:-round3}
O=gpurun_out/${TAG}p
mkdir -p $O
python bench.py > $O/${TAG}_bench_n1.json 2> $O/bench_n1.err
python bench.py --dtype bf16 --no-cpu-baseline > $O/${TAG}_bench_n1_bf16.json 2>/dev/null
python bench.py --batch 2 --no-roofline --no-cpu-baseline > $O/${TAG}_bench_b2.json 2>/dev/null
python bench.py --batch 2 --graph --no-cpu-baseline > $O/${TAG}_bench_b2_graph.json 2>/dev/null
python bench.py --config 1 --no-cpu-baseline > $O/${TAG}_bench_config1.json 2>/dev/null
python bench.py --config 1 --graph --no-cpu-baseline > $O/${TAG}_bench_config1_graph.json 2>/dev/null
python bench.py --config 3 --no-cpu-baseline > $O/${TAG}_bench_config3.json 2>/dev/null
python bench.py --config 4 --no-cpu-baseline > $O/${TAG}_bench_config4.json 2>/dev/null
python bench.py --config 5 --no-cpu-baseline > $O/${TAG}_bench_config5.json 2>/dev/null
RUNET_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline > $O/${TAG}_bench_n1_rccl_one_rank.json 2>/dev/null
RUNET_BENCH_FORCE_DIST=1 python bench.py --config 5 --no-cpu-baseline > $O/${TAG}_bench_config5_rccl_one_rank.json 2>/dev/null
echo benches done
bash tools/collect_profiles.sh $O ${TAG}
cp $O/${TAG}_pmc_traffic.json $O/${TAG}_pmc_traffic_runet_f32_16x256.json
echo fp32 profile done
bash tools/collect_profiles.sh $O ${TAG}_bf16 --dtype bf16
cp $O/${TAG}_bf16_pmc_traffic.json $O/${TAG}_pmc_traffic_runet_bf16_16x256.json
bash tools/collect_profiles.sh $O ${TAG}_config3 --config 3
cp $O/${TAG}_config3_pmc_traffic.json $O/${TAG}_pmc_traffic_runet_bf16_4x512.json
echo reduced-precision profiles done
ROOT=$(pwd)
( cd /tmp && export TMPDIR=/tmp && RUNET_NO_WGRAD_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/ss_trace -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1 )
python3 tools/prof_summary.py $O/ss_trace 13 > $O/${TAG}_kernel_stats_single_stream.txt
cp $(ls $O/ss_trace/*/*kernel_stats.csv | head -1) $O/${TAG}_kernel_stats_single_stream.csv
python3 tools/step_sequence.py $O/ss_trace > $O/${TAG}_step_sequence.txt
rm -rf $O/ss_trace
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $ROOT/$O/gap_trace -- python3 $ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2>&1 )
python3 tools/trace_gaps.py $O/gap_trace 0.5 > $O/${TAG}_trace_gaps.txt
rm -rf $O/gap_trace
( cd /tmp && export TMPDIR=/tmp && RUNET_NO_WGRAD_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/b2_trace -- python3 $ROOT/bench.py --batch 2 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1 )
python3 tools/prof_summary.py $O/b2_trace 13 > $O/${TAG}_kernel_stats_batch2.txt
python3 tools/step_sequence.py $O/b2_trace | head -1 >> $O/${TAG}_kernel_stats_batch2.txt
rm -rf $O/b2_trace
echo traces done
python tools/conv_launches.py > $O/${TAG}_conv_launches.txt 2>/dev/null
python tools/conv_launches.py --dtype bf16 > $O/${TAG}_bf16_conv_launches.txt 2>/dev/null
python tools/bench_gemm_x3.py > $O/${TAG}_gemm_x3.txt 2>/dev/null
echo all done
