#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 120 ./tools/probe > gpurun_out/probe.log 2>&1 || { echo probe failed; tail gpurun_out/probe.log; exit 1; }
cat gpurun_out/probe.log
timeout -k 10 300 ./tools/stream_tune > gpurun_out/tune2.log 2>&1 || { echo tune failed; tail gpurun_out/tune2.log; exit 1; }
tail -5 gpurun_out/tune2.log
