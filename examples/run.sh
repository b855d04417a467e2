#!/bin/bash
# reference examples/run.sh: run from the repository root on a machine with an MI355X
set -e
export PYTHONPATH="$(pwd):$(pwd)/examples:$PYTHONPATH"
python examples/make_data.py
python examples/example1.py
python examples/example2.py
python examples/example3.py
python examples/example4.py
