# per-shard pass time at the 1/2/4/8-GPU shard sizes of the N = 10 M bench: plain, and through the all-gather + merge path (world 1)
cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29544 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for rows in 10000000 5000000 2500000 1250000; do
a=$(python bench.py --rows $rows --steps 30 --warmup 3 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
b=$(python bench.py --rows $rows --steps 30 --warmup 3 --no-cpu-baseline --no-extras --force-dist 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
echo "rows $rows  ms_per_pass $a  with_gather_and_merge $b"
done
