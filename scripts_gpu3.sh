timeout 900 python -m pytest tests -m gpu -q -k vertices_to_faces 2>&1 | grep -E "Error|error|assert|^E" | head -12
