#!/bin/bash
GPU_MAX_HW_QUEUES=8 python tools/two_contexts_probe.py
for plain in 0 1; do S2K_SUBMIT_PLAIN=$plain python tools/pipeline_trace.py 24; done
for plain in 0 1; do S2K_SUBMIT_PLAIN=$plain python tools/pipeline_trace.py 24; done
