#!/bin/bash
# the table of DESIGN 4a: step time by signatures per key, grouping off / auto / always
timeout 600 python tools/keyed_probe.py 20 0,10,16,17,18,20 2>/dev/null
