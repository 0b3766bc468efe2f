# Round 6: kernel traces of the two pixel configurations + the online actor update rows (GPU box)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6/pix; mkdir -p $O
python3 tools/bench_pixels.py dmc 5 > /dev/null 2>&1   # (a fresh box runs ~12 % slow at first)
for c in dmc atari; do
  python3 tools/bench_pixels.py $c 30 > $O/${c}_plain.txt 2>&1; tail -1 $O/${c}_plain.txt
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$c -o t -- python3 tools/bench_pixels.py $c 20 > $O/kt_$c.log 2>&1
  python3 tools/rocpd_summary.py $O/kt_$c/t_results.db 26 > $O/trace_$c.md; head -32 $O/trace_$c.md
  rm -rf $O/kt_$c
done
python3 -c "
import sys, json; sys.path.insert(0,'.')
import torch, bench
print(json.dumps(bench.actor_update_rows(torch.device('cuda'))))" > $O/actor_rows.json 2>$O/actor_rows.err; cat $O/actor_rows.json; tail -2 $O/actor_rows.err
