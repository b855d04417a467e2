python scripts/glue_profile.py 2>&1 | grep -v "^-" | cut -c1-72,100-175 | grep -A13 "=====" | head -40
