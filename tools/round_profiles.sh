set -u
bash tools/make_profile.sh atrium atrium > gpurun_out/prof_atrium.log 2>&1 || { tail gpurun_out/prof_atrium.log; exit 1; }
PROFILE_ARGS="--rb --spp 16 --steps 32 --warmup 2" bash tools/make_profile.sh atrium atrium_rb16 > gpurun_out/prof_rb16.log 2>&1 || { tail gpurun_out/prof_rb16.log; exit 1; }
bash tools/make_profile.sh car car > gpurun_out/prof_car.log 2>&1 || { tail gpurun_out/prof_car.log; exit 1; }
for d in atrium atrium_rb16 car; do for sub in sq1 sq2 tcc fetch write tcp; do rm -rf gpurun_out/profile_$d/$sub; done; f=$(find gpurun_out/profile_$d/trace -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/profile_$d/kernel_stats.csv; rm -rf gpurun_out/profile_$d/trace; done
echo profiles done
