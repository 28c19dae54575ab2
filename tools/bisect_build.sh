#!/bin/bash
# build library variants with -mllvm -opt-bisect-limit=N for each N given
for n in "$@"; do
  (make -C pwnfps_amd/csrc VARIANT=bis$n EXTRA="-mllvm -opt-bisect-limit=$n" > /tmp/build_bis$n.log 2>&1; echo bis$n $(grep -c "rror" /tmp/build_bis$n.log)) &
done
wait
