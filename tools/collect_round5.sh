# after `gpurun -- bash tools/snapshot_round5.sh r05`: copy what the docs cite from gpurun_out/r05 into profiles/ (names demangled)
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r05}
P=${2:-r05}
cp $O/bench.json profiles/${P}_bench_default.json
cp $O/bench_records.jsonl profiles/${P}_bench_default_records.jsonl
cp $O/bench_launcher_w1.json profiles/${P}_bench_launcher_w1.json
cp $O/bench_shard_125k.json profiles/${P}_bench_shard_125k.json
cp $O/bench_rawnet2.json profiles/${P}_bench_rawnet2.json
for d in ecapa_bf16 rawnet2_f16 ecapa_f32 ecapa_f32x3 rawnet2_f32x3 scoring; do
  f=$(find $O/prof/$d -name "*kernel_stats.csv" | head -1)
  python tools/demangle.py < $f > profiles/${P}_${d}_kernel_stats.csv
done
python tools/pmc_summary3.py $O/pmc profiles/${P}_pmc_summary.json profiles/pmc_traffic.json > profiles/${P}_pmc_table.txt
cp $O/scoring_pmc/table.txt profiles/${P}_scoring_pmc_table.txt
