#!/bin/bash
# time experiment builds (make -C pwnfps_amd/csrc VARIANT=tag ...) of the trace kernel
#   tools/variants.sh "tag1 tag2 ..." [W H [FRAMES [LEVEL [BLUR]]]]
TAGS=$1; shift
export PWN_HASH=1
for t in $TAGS; do
	if [ "$t" = base ]; then unset PWNHIP_LIB; else export PWNHIP_LIB=$PWD/pwnfps_amd/libpwnhip_$t.so; fi
	python3 tools/prof_frame.py "$@" 2>&1 | tail -${PWN_TAIL:-1}
done
