timeout 900 python -m pytest tests -m gpu -q -k "examples" 2>&1 | tail -15
