#!/bin/bash
# Throw-away build of the library with ONE kernel file replaced by a patched copy (for same-box A/B timing; never shipped):
#   tools/variant_lib.sh <name> <file.hip> <python-expression that maps the source text `s` to the patched text>
# -> historian_amd/lib_w<name>/libhistorian_hip.so (git-ignored; selected with HX_LIB_PATH).  The other objects are the
# product build's (historian_amd/build).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; file=$2; expr=$3
src=$R/historian_amd/csrc
tmp=$(mktemp -d)
python3 - "$src/$file" "$tmp/$file" "$expr" <<'PY'
import sys
s = open(sys.argv[1]).read()
t = eval(sys.argv[3], {"s": s})
assert t != s, "the patch changed nothing"
open(sys.argv[2], "w").write(t)
PY
mkdir -p $R/historian_amd/lib_w$name
obj=$tmp/${file%.hip}.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -I$src -I$R/include -c -o $obj $tmp/$file
objs=$(ls $R/historian_amd/build/*.o | grep -v "/${file%.hip}.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/historian_amd/lib_w$name/libhistorian_hip.so $objs $obj
rm -rf $tmp
echo "built historian_amd/lib_w$name/libhistorian_hip.so"
