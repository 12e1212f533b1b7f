# FETCH_SIZE factors of the two 19-plane calibration kernels (aligned / shifted windows): bash tools/dev/calib_shift.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/calibs
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/calibs -o x -- python -c "
import sys; sys.path.insert(0, '$R')
import bench; print(bench.measured_hbm(0))" > /dev/null 2>&1
python $R/tools/rocprof_summary.py $(find $R/gpurun_out/calibs -name x_results.db | head -1) --pmc | grep -E "calib_(pull|shift)19.*FETCH" | cut -c1-40,88-140
rm -rf $R/gpurun_out/calibs
echo "true bytes per launch: $((19 * 48 * 1024)) KB"
