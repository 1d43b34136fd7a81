#!/bin/bash
cd $GRAFT_REPO_ROOT
L=$PWD/basic_pitch_amd/lib
for i in 1 2; do
echo "== default"; tools/ab_run.sh | tail -1
for v in nc2 nc3 nc6 cmc3 cmc5 cmc6 cmpf2 cmpf0; do echo "== $v"; BASIC_PITCH_AMD_LIB=$L/var_$v.so tools/ab_run.sh | tail -1; done
done
