set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cli-wall --dev-skip-fits > gpurun_out/exp1/skipfits.json 2> gpurun_out/exp1/skipfits.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cli-wall --dev-skip-fits --hist-on-main > gpurun_out/exp1/skipfits_histmain.json 2> gpurun_out/exp1/skipfits_histmain.err
echo done
