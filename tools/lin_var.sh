line() { tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print(l['roofline']['avg_launch_us'])"; }
python bench.py --workload cfg3 --no-cpu-baseline --steps 100 | line
for i in 1 2 3; do python bench.py --workload linearitystd --steps 100 | line; done
for i in 1 2; do python bench.py --workload linearitystd --steps 100 --prewarm-s 3 | line; done
for i in 1 2; do python bench.py --workload linearitystd --steps 100 --no-cpu-baseline | line; done
