#!/bin/bash
timeout 600 python -m pytest tests/test_c_harness.py -m gpu -q 2>&1 | tail -3
