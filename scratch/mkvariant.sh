#!/bin/bash
# build a measurement variant of the library (with the test hooks, like lib/hooks/): scratch/mkvariant.sh NAME "-DFOO -DBAR"  ->  scratch/prev_lib/libaidax_NAME.so (AIDAX_LIB selects it)
set -e
name=$1; shift
make -j8 OBJDIR=build/obj_$name LIBDIR=build/lib_$name EXTRA="-DAIDAX_TEST_HOOKS $*" build/lib_$name/libaidax_hip.so 2>&1 | grep -E "error|warning: v|spill" || true
mkdir -p scratch/prev_lib
cp build/lib_$name/libaidax_hip.so scratch/prev_lib/libaidax_$name.so
ls -la scratch/prev_lib/libaidax_$name.so
