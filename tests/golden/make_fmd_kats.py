"""Regenerates fmd_kats.json's `issue39.text` from the reference source (the 2.7 kb literal of
fmindex.rs:808-858); the other entries are transcribed by hand with their file:line.
Usage (in a container that has /root/reference): python tests/golden/make_fmd_kats.py"""
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
src = open("/root/reference/src/data_structures/fmindex.rs").read()
a = src.index('let reads = b"', src.index("fn test_issue39")) + len('let reads = b"')
reads = re.sub(r"\\\n\s*", "", src[a:src.index('";', a)])
doc = json.load(open(os.path.join(HERE, "fmd_kats.json")))
doc["issue39"]["text"] = reads
json.dump(doc, open(os.path.join(HERE, "fmd_kats.json"), "w"), indent=1)
