#!/bin/bash
timeout 600 python -m pytest "tests/test_hip_parity.py::test_teapot_views_backward" -m gpu -q --tb=long -x -p no:cacheprovider 2>&1 | tail -60
