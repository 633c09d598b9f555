# after `gpurun -- bash tools/snapshot_round4.sh r04`: copy what the docs cite from gpurun_out/r04 into profiles/ (names demangled)
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r04}
cp $O/bench.json profiles/r04_bench_default.json
cp $O/bench_launcher_w1.json profiles/r04_bench_launcher_w1.json
cp $O/bench_shard_125k.json profiles/r04_bench_shard_125k.json
cp $O/bench_rawnet2.json profiles/r04_bench_rawnet2.json
for d in ecapa_bf16 rawnet2_f16 ecapa_f32 ecapa_f32x3 rawnet2_f32x3 scoring; do
  f=$(find $O/prof/$d -name "*kernel_stats.csv" | head -1)
  python tools/demangle.py < $f > profiles/r04_${d}_kernel_stats.csv
done
python tools/pmc_summary3.py $O/pmc profiles/r04_pmc_summary.json profiles/pmc_traffic.json > profiles/r04_pmc_table.txt
