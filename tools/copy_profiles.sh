#!/bin/bash
# copies what tools/refresh_profiles.sh left under gpurun_out/refresh/ into profiles/ under the round's names: copy_profiles.sh r04
set -eu
R=${1:?round tag, e.g. r04}
S=gpurun_out/refresh; D=profiles
cp $S/bench.json $D/${R}_bench_c2_n1.json
cp $S/bench_bf16x3.json $D/${R}_bench_c2_n1_bf16x3_optin.json
cp $S/kernel_stats.csv $D/${R}_kernel_stats_bench_c2.csv
cp $S/pmc_sim.json $D/${R}_pmc_sim.json
cp $S/sim_prof.txt $D/${R}_sim_prof.txt
cp $S/configs.json $D/${R}_all_configs_n1.json
cp $S/configs_threads1.json $D/${R}_all_configs_n1_one_host_thread.json
cp $S/configs_one_rng_stream.json $D/${R}_all_configs_n1_one_rng_stream.json
for k in c3 c4 c5 w9x128 w19x64; do cp $S/kernel_stats_$k.csv $D/${R}_kernel_stats_$k.csv; cp $S/pmc_$k.json $D/${R}_pmc_$k.json; done
cp $S/c5_move_timeline.txt $D/${R}_c5_move_timeline.txt
cp $S/sim_prof_c5_rounds.txt $D/${R}_sim_prof_c5_rounds.txt
cp $S/sim_prof_c5_no_rounds.txt $D/${R}_sim_prof_c5_no_rounds.txt
cp $S/c5_450moves.json $D/${R}_c5_450_moves.json
cp $S/c5_450moves_threads1.json $D/${R}_c5_450_moves_one_host_thread.json
cp $S/c5_no_rounds.json $D/${R}_c5_without_rounds.json
cp $S/c5_round3_path.json $D/${R}_c5_round3_path_one_workgroup_per_leaf.json
cp $S/c5_no_pairs.json $D/${R}_c5_without_pairs.json
cp $S/wide_configs.json $D/${R}_wide_configs_n1.json
cp $S/c5_512_games_one_gpu.json $D/${R}_c5_512_games_one_gpu_not_the_baseline_shard.json
cp $S/time_wide_towers.json $D/${R}_time_wide_towers.json
for k in w9x128 w19x64; do cp $S/sim_prof_$k.txt $D/${R}_sim_prof_$k.txt; done
ls $D/${R}_* | wc -l
