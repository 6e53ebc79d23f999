#!/bin/bash
# tools/g.sh <timeout-seconds> <logname> '<command>' -- rebuild everything that travels, then run the
# command on the GPU box in the background; output in gpurun_out/<logname>.log
T=$1; L=$2; shift 2
make -s -C sdrreceiver_amd/csrc && make -s -C host && make -s -C oracle liborc.so && make -s -C oracle/ref && make -s -C host/qt || { echo BUILD FAILED; exit 1; }
mkdir -p gpurun_out
# the GPU box gets no .git: leave the build's identity where tools/profile.sh can pick it up
echo "$(git rev-parse HEAD)$(git diff --quiet HEAD -- sdrreceiver_amd include || echo '+dirty:'$(git diff HEAD -- sdrreceiver_amd include | sha1sum | cut -c1-10))" > profiles/build_sha.txt
nohup /usr/local/graft/bin/gpurun --timeout $T -- "$*" > gpurun_out/$L.log 2>&1 &
echo "started $L"
