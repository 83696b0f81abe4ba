#!/bin/bash
# On a GPU box, with pathtracing_amd/libptamd_clk.so built by `tools/build_variant.sh clk pt_exact -DPT_PKT_CLOCKS`:
# share of the search kernel's wave clocks per part of the packet walk (atrium, -rb, mainOMP model), and events per packet.
cp pathtracing_amd/libptamd.so /tmp/libptamd_base.so && cp pathtracing_amd/libptamd_clk.so pathtracing_amd/libptamd.so
timeout -k 10 300 python3 bench.py ${PKT_ARGS:-} --rb --spp 16 --steps 16 --warmup 2 --no-cpu-baseline --no-rm-leg 2> gpurun_out/pkt_clk.err > /dev/null || { tail gpurun_out/pkt_clk.err; cp /tmp/libptamd_base.so pathtracing_amd/libptamd.so; exit 1; }
cp /tmp/libptamd_base.so pathtracing_amd/libptamd.so
python3 - <<'PY'
import re
names = ["pop", "node fetch", "packet test", "triangle fetch+test", "step to children", "packet set-up/results", "first packet test of a packet", "(the first packet of a wave, whole)"]
acc = [0] * 8; cnt = [0] * 8; n = 0
for line in open("gpurun_out/pkt_clk.err"):
    if not line.startswith("pkt_clk"): continue
    n += 1
    for k, f in enumerate(line.split()[1:]):
        a, c = f.split("/"); acc[k] += int(a); cnt[k] += int(c)
tot = sum(acc[:7])
print(f"{n} launches; packets per launch {cnt[5] / n / 2:.0f}")
for k in range(8):
    print(f"{names[k]:24s} {100 * acc[k] / tot:5.1f} %   {cnt[k] / (cnt[5] / 2):7.1f} per packet   {acc[k] / max(cnt[k], 1):7.0f} clocks each")
PY
