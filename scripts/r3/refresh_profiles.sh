#!/bin/bash
# Run from the repo root in the build container: profiles the three closest-hit workloads and the two path-tracing
# renders on the GPU box (one gpurun call) and, only if every pass came back, rewrites profiles/r3_*.
set -e
cd "$(dirname "$0")/../.."
rm -rf gpurun_out/prof_r3_cornell gpurun_out/prof_r3_mesh1m gpurun_out/prof_r3_fog gpurun_out/ptprof_r3pt gpurun_out/ptprof_r3fog
/usr/local/graft/bin/gpurun --timeout 1100 -- 'bash scripts/profile_gpu.sh r3_cornell 5 "--workload cornell" > gpurun_out/p_cornell.log 2>&1 && bash scripts/profile_gpu.sh r3_mesh1m 5 "--workload mesh1m" > gpurun_out/p_mesh1m.log 2>&1 && bash scripts/profile_gpu.sh r3_fog 5 "--workload fog" > gpurun_out/p_fog.log 2>&1 && bash scripts/profile_pt.sh r3pt Config_Glossy1024.xml > gpurun_out/p_pt.log 2>&1 && bash scripts/profile_pt.sh r3fog Config_Fog1024.xml > gpurun_out/p_ptfog.log 2>&1' 2>&1 | tail -2
for d in prof_r3_cornell prof_r3_mesh1m prof_r3_fog ptprof_r3pt ptprof_r3fog; do
  ls gpurun_out/$d/grbm/*/*_counter_collection.csv > /dev/null
  ls gpurun_out/$d/kt/*/*_kernel_stats.csv > /dev/null
done
for w in cornell mesh1m fog; do
  python3 scripts/summarize_profile.py gpurun_out/prof_r3_$w profiles/r3_${w}_traffic.json > profiles/r3_${w}_summary.txt
  cp "$(ls -t gpurun_out/prof_r3_$w/kt/*/*_kernel_stats.csv | head -1)" profiles/r3_${w}_kernel_stats.csv
done
python3 scripts/summarize_profile.py gpurun_out/ptprof_r3pt profiles/r3_pt_cornell_traffic.json > profiles/r3_pt_cornell_summary.txt
cp "$(ls -t gpurun_out/ptprof_r3pt/kt/*/*_kernel_stats.csv | head -1)" profiles/r3_pt_cornell_kernel_stats.csv
python3 scripts/summarize_profile.py gpurun_out/ptprof_r3fog profiles/r3_pt_fog_traffic.json > profiles/r3_pt_fog_summary.txt
cp "$(ls -t gpurun_out/ptprof_r3fog/kt/*/*_kernel_stats.csv | head -1)" profiles/r3_pt_fog_kernel_stats.csv
echo "profiles refreshed"
