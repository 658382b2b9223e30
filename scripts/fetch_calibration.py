#!/usr/bin/env python3
"""Known-bytes kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 in THIS library's access patterns:
run under `rocprofv3 --pmc FETCH_SIZE -- python3 scripts/fetch_calibration.py` (and once more with WRITE_SIZE); every
kernel below moves a byte count that is known exactly, so FETCH_SIZE * 1024 / known = the counter's factor for that
pattern.  Prints the known bytes per launch as JSON; scripts/summarize_profile.py joins them with the counter CSV.
  k_gather_records<READ, REC>   lane i reads READ x 16 bytes of record perm(i) (REC x 16 bytes each), every record once
  k_field_streams<false>        38 coalesced 8-byte streams in, 13 out (the pinned path's per-substep traffic)
  k_copy16_flat<false>          16 bytes per lane, coalesced (the guide's own calibration case)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from constraint_solver_amd import capi  # noqa: E402

RECORDS = 1 << 22
PATTERNS = [(128, 128), (128, 64), (128, 16), (192, 192), (192, 64), (256, 128)]   # (record bytes, bytes read)


def main():
    known = {}
    for record_bytes, read_bytes in PATTERNS:
        rate = capi.selftest_gather(RECORDS, record_bytes, read_bytes, repeats=3)
        known["k_gather_records<%du, %du>" % (read_bytes // 16, record_bytes // 16)] = {
            "read_bytes_requested": RECORDS * read_bytes, "write_bytes": RECORDS * 8, "records": RECORDS, "record_bytes": record_bytes,
            "launches": 4, "gbytes_per_s": rate}
    bodies = 1 << 21
    rate = capi.selftest_field_streams(bodies, tile_major=False, repeats=3)
    known["k_field_streams<false>"] = {"read_bytes_requested": bodies * 38 * 8, "write_bytes": bodies * 13 * 8, "launches": 4, "gbytes_per_s": rate}
    rate = capi.selftest_hbm_copy(1 << 30, 3)
    known["k_copy16_flat<false>"] = {"read_bytes_requested": 1 << 30, "write_bytes": 1 << 30, "launches": 4, "gbytes_per_s": rate}
    print(json.dumps(known))


if __name__ == "__main__":
    main()
