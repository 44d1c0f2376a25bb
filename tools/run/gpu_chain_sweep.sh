#!/bin/bash
# per-kind chain kernel time (us per launch) over batch size x tile policy -> gpurun_out/chain_sweep/table.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/chain_sweep; mkdir -p $O; : > $O/table.txt
for B in ${BATCHES:-1 2 4 8 12 16 24 32}; do
  for T in ${TILES:-1 2 3}; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o s -- python tools/small_batch.py --batches $B --steps 10 --warmup 3 --option chain_tile=$T > /dev/null 2> $O/err.txt
    F=$(find $O/p -name "*kernel_stats.csv" | head -1)
    python - "$F" $B $T >> $O/table.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = {}
for r in rows:
    n = r["Name"]
    if "chain" in n and "kernel<" in n:
        kind = n.split("<")[1].split(">")[0].split(",")
        k = int(kind[3]) if "chain_kernel" in n else int(kind[2])
        out[k] = out.get(k, 0) + float(r["TotalDurationNs"]) / 1e3
calls = {0: 6, 1: 3, 2: 3, 3: 2}
n_fw = 13 + 1   # warmup + steps + the first call
print(f"B {int(sys.argv[2]):3d} tile {sys.argv[3]}  " + "  ".join(f"kind{k} {out.get(k, 0) / (calls[k] * n_fw):7.1f} us" for k in range(4))
      + f"   total/forward {sum(out.values()) / n_fw / 1e3:6.3f} ms")
PY
    rm -rf $O/p
  done
done
cat $O/table.txt
