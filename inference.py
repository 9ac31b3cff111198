#!/usr/bin/env python3
"""Drop-in for the reference's `python inference.py --tgt <image|folder> ...` (same flags) on the MI355X engine."""
from callireader_amd.inference import main

if __name__ == '__main__':
    main()
