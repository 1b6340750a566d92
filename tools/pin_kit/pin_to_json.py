#!/usr/bin/env python3
"""`PIN <key> <f32 bits, hex> <decimal>` lines (what tools/pin_kit/pin_kit.rs prints inside the reference tree) -> tests/golden/reference_pin.json.
usage: python tools/pin_kit/pin_to_json.py pin.txt > tests/golden/reference_pin.json"""
import json
import struct
import sys

out = {}
for line in open(sys.argv[1]):
    parts = line.split()
    if len(parts) >= 3 and parts[0] == "PIN":
        out[parts[1]] = {"bits": parts[2], "value": struct.unpack("<f", struct.pack("<I", int(parts[2], 16)))[0]}
json.dump({"source": "printed by tools/pin_kit/pin_kit.rs inside gillett-hernandez/rust-pathtracer (cargo test pin_kit -- --nocapture)", "pins": out}, sys.stdout, indent=1, sort_keys=True)
print()
