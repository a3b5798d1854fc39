#!/bin/sh
# TEST INFRASTRUCTURE - builds oracle/_ref/libptref.so, the reference's own render core compiled
# as a shared library, from the sources where they lie under /root/reference.
#
# Nothing of the reference is copied into the repository: the line ranges the render core lives
# in are cut into a temporary directory, compiled, and the directory is removed.  The only output
# is the shared library under oracle/_ref/ (git-ignored).  When the reference tree is absent
# (GPU box) the script does nothing and the prebuilt library, if any, is used as is.
#
# Dialect accommodations (no arithmetic is touched):
#   #define ushort ref_ushort    (in ref_harness.cpp) rt_types.cpp:8 typedefs `ushort`, which glibc also declares
#   -D__declspec(x)=__thread     rt_misc.cpp:1163 `__declspec(thread)` (MSVC keyword)
#   rt_math.cpp:1100,1109        union Vec4 names `w` and `a` twice (an MSVC extension); renamed
# Parameterisations of literals the reference hard-codes (needed by the test configurations):
#   app.cpp:1276  grid hint vec3i(10,10,10)  -> vec3i(ref_grid_hint, ...)
# "fixed" variant only (the verbatim variant is compiled too):
#   app.cpp:1331  relocated cell list goes to the tail of the shared buffer, not its base
#   app.cpp:1286  buffer sized up front so the overflow/double-free path is never entered
set -e
REF=${PT_REFERENCE:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT="$HERE/_ref"
if [ ! -f "$REF/code/raycast.cpp" ]; then
	echo "ref_build: no reference tree at $REF, keeping prebuilt oracle/_ref as is"
	exit 0
fi
mkdir -p "$OUT"
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
C="$REF/code"

sed -n '6,33p'       "$C/rt_misc.cpp" > "$TMP/misc_macros.inc"
sed -n '86,92p'      "$C/rt_misc.cpp" > "$TMP/misc_memcpy.inc"
sed -n '1150,1251p'  "$C/rt_misc.cpp" > "$TMP/misc_pcg.inc"
sed -n '1255,1383p'  "$C/rt_misc.cpp" > "$TMP/misc_darray.inc"
sed -e '1100s/float w;/float w_dup;/' -e '1109s/float a;/float a_dup;/' "$C/rt_math.cpp" > "$TMP/math_patched.inc"
sed -n '3,1356p'     "$C/raycast.cpp" > "$TMP/rc_core.inc"
sed -n '1888,1906p'  "$C/raycast.cpp" > "$TMP/rc_defobj.inc"
sed -n '2064,2108p'  "$C/raycast.cpp" > "$TMP/rc_precalc.inc"
sed -n '669,727p'    "$C/app.cpp"     > "$TMP/app_tables.inc"
sed -n '1075,1079p;1089,1090p' "$C/app.cpp" > "$TMP/app_camera.inc"
sed -e '1276s/vec3i(10,10,10)/vec3i(ref_grid_hint,ref_grid_hint,ref_grid_hint)/' "$C/app.cpp" \
	| sed -n '1161,1357p' > "$TMP/app_setup_verbatim.inc"
sed -e '1276s/vec3i(10,10,10)/vec3i(ref_grid_hint,ref_grid_hint,ref_grid_hint)/' \
    -e '1286s/cellCount\*10/ref_grid_buffer_bound(world, grid)/' \
    -e '1331s/settings->gridBuffer;/settings->gridBuffer + settings->gridBufferCount;/' "$C/app.cpp" \
	| sed -n '1161,1357p' > "$TMP/app_setup_fixed.inc"
sed -n '1359,1419p'  "$C/app.cpp"     > "$TMP/app_jobs.inc"
sed -n '1578,1604p'  "$C/raycast.cpp" > "$TMP/rc_scenefile.inc"
sed -n '351,353p;3634,3640p' "$C/rt_math.cpp" > /dev/null   # (present in math_patched.inc; listed for the reader)

g++ -std=gnu++14 -O2 -fpermissive -w -ffp-contract=off -fPIC -shared \
	'-D__declspec(x)=__thread' \
	-I"$TMP" -I"$C" -I"$HERE/../include" \
	"$HERE/ref_harness.cpp" -o "$OUT/libptref.so" -lpthread
echo "ref_build: built $OUT/libptref.so"
