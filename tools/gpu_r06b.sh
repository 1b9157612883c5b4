#!/bin/bash
export TMPDIR=/tmp
python tools/share_ab.py --reserved 0 129 124 123 122 0 129
python tools/share_ab.py --lib variants/lib_base.so --reserved 0 0
