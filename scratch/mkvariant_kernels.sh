#!/bin/bash
# A measurement variant that differs from the tree's build only in aidax_kernels.hip's -D switches: the other objects are copied from
# build/obj_hooks, aidax_kernels.hip recompiled.   scratch/mkvariant_lp.sh NAME "-DFOO=1 -DBAR"  ->  scratch/prev_lib/libaidax_NAME.so
set -e
name=$1; shift
make -s -j8 all > /dev/null
rm -rf build/obj_$name; cp -a build/obj_hooks build/obj_$name
rm -f build/obj_$name/aidax_kernels.o build/obj_$name/aidax_kernels.remarks build/obj_$name/scratch.ok
make -j8 OBJDIR=build/obj_$name LIBDIR=build/lib_$name EXTRA="-DAIDAX_TEST_HOOKS $*" build/lib_$name/libaidax_hip.so 2>&1 | grep -E "error|warning: v|spill|scratch" || true
mkdir -p scratch/prev_lib
cp build/lib_$name/libaidax_hip.so scratch/prev_lib/libaidax_$name.so
ls -la scratch/prev_lib/libaidax_$name.so
