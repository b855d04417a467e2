timeout 900 python -m pytest tests -m gpu -q -k "public_api or none_grad" 2>&1 | tail -12
