#!/bin/bash
# Device assembly of every csrc/*.hip file that changed since a commit, old tree vs working tree, compared kernel by kernel
# (tools/isa_diff.py): which instantiations of the DEFAULT path are not the binaries that ran at that commit.  No GPU needed.
#   tools/isa_diff_since.sh 298c878 > profiles/r04_isa_diff_since_298c878.txt
set -e
BASE=${1:?commit}; R=$(cd "$(dirname "$0")/.." && pwd); W=$(mktemp -d)
# first line: what the file was made from -- tests/test_host_logic.py fails when the kernel sources have changed since
echo "# csrc digest $(python3 $R/tools/csrc_digest.py) | base $BASE | $(/opt/rocm/bin/hipcc --version | grep -m1 -o 'HIP version.*')"
mkdir -p $W/old/csrc $W/old/include $W/new
git -C $R archive $BASE speech2affective_gestures_amd/csrc include | tar -x -C $W/old --strip-components=0
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only"
# a changed header reaches every translation unit: then ALL kernel files are compared, not only the edited ones
if git -C $R diff --name-only $BASE -- speech2affective_gestures_amd/csrc include | grep -q '\.h$'; then
  FILES=$(cd $R && ls speech2affective_gestures_amd/csrc/*.hip)
else
  FILES=$(git -C $R diff --name-only $BASE -- speech2affective_gestures_amd/csrc | grep '\.hip$')
fi
for f in $FILES; do
  b=$(basename $f .hip)
  [ -f $W/old/$f ] || { echo "== $b: new file"; continue; }
  /opt/rocm/bin/hipcc $FLAGS -I$W/old/include -I$W/old/speech2affective_gestures_amd/csrc $W/old/$f -o $W/old/$b.s 2>/dev/null
  /opt/rocm/bin/hipcc $FLAGS -I$R/include -I$R/speech2affective_gestures_amd/csrc $R/$f -o $W/new/$b.s 2>/dev/null
  echo "== $b"
  python3 $R/tools/isa_diff.py $W/old/$b.s $W/new/$b.s --brief || true
done
rm -rf $W
