#!/usr/bin/env python3
"""Config 3 (DecayingTurbulence3D-shaped 512^3 periodic, RK44 + spectral) step loop for rocprofv3: tools/turb_prof.py [n] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import configs_sanity as c
c.run_turb(int(sys.argv[1]) if len(sys.argv) > 1 else 512)
