#!/bin/bash
# host-buffer entry point, grouping off / on
timeout 300 python tools/host_probe.py 2>/dev/null
