#!/bin/bash
timeout 600 python -m pytest "tests/test_hip_parity.py::test_teapot_views_backward" -m gpu -q --tb=short -x -p no:cacheprovider 2>&1 | grep -E "Error|assert|passed|failed" | head -12
