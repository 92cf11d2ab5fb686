#!/bin/bash
python tools/pageable_probe.py notorch 8 | cut -c1-700
python tools/pageable_probe.py torch 8 | cut -c1-700
python tools/pageable_probe.py torch 16 | cut -c1-700
