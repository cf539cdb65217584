# usage: bash scripts/gpu_hipstats.sh <tag> <n_lines> <command...>   -> HIP API stats (rocprofv3 --hip-trace --stats) of the command
tag=$1; n=$2; shift 2
root=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/hprof_$tag
timeout 900 rocprofv3 --hip-trace --stats --output-format csv -d /tmp/hprof_$tag -o $tag -- "$@" > /tmp/hprof_$tag.log 2>&1
f=$(find /tmp/hprof_$tag -name "*hip_api_stats.csv" | head -1)
if [ -n "$f" ]; then cp $f $root/gpurun_out/${tag}_hip_api_stats.csv; head -$n $f | cut -d, -f1-6; else tail -20 /tmp/hprof_$tag.log; fi
grep playlists /tmp/hprof_$tag.log | cut -c1-80
cd $root
