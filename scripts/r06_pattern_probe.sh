#!/bin/bash
# Which pattern libraries does the GPU suite ask for that the tree does not hold?  Runs the suite with the pattern cache in a scratch
# directory seeded with the tree's libraries; what appears there besides is compiled on the box — on EVERY fresh box, 40 s apiece.
cd "$(dirname "$0")/.."
export LQG_PAT_DIR=$PWD/gpurun_out/pat_probe
rm -rf $LQG_PAT_DIR; mkdir -p $LQG_PAT_DIR
cp lqg_amd/csrc/pat/*.so lqg_amd/csrc/pat/*.stamp $LQG_PAT_DIR/
ls $LQG_PAT_DIR/*.so | sort > gpurun_out/pat_probe_before.txt
python -m pytest tests/ -q -m gpu -x 2>&1 | tail -3
ls $LQG_PAT_DIR/*.so | sort > gpurun_out/pat_probe_after.txt
comm -13 gpurun_out/pat_probe_before.txt gpurun_out/pat_probe_after.txt
# keep only the sources of the new ones for the merge back (the .so are large)
for f in $(comm -13 gpurun_out/pat_probe_before.txt gpurun_out/pat_probe_after.txt); do echo "NEW $(basename $f)"; done
find $LQG_PAT_DIR -name "*.so" -delete; find $LQG_PAT_DIR -name "*.o" -delete
