rm -rf gpurun_out/*
bash tools/profile_round.sh r02a > /tmp/prof_r02a.log 2>&1; tail -30 /tmp/prof_r02a.log
for f in gpurun_out/prof_r02a/*.err; do echo "== $f"; tail -n 6 $f; done
mkdir -p gpurun_out/profiles_r02a; cp profiles/r02a_* profiles/traffic_latest.json gpurun_out/profiles_r02a/ 2>/dev/null
du -sh gpurun_out/prof_r02a/*; rm -rf gpurun_out/prof_r02a/{stats,fetch,write,sq1,sq2,sw1,sw2}
