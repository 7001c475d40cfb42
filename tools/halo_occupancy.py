#!/usr/bin/env python3
"""Workgroups per CU of the LDS-halo conv kernels at a range of dynamic-LDS sizes (gca_conv_halo_occupancy)."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H = importlib.import_module('video-graph-ssl_amd')._hip

for tm, tn, m in [(2, 2, 2), (2, 2, 0), (5, 1, 2), (4, 1, 2), (3, 1, 2), (2, 1, 2)]:
    print((tm, tn, m), [(lds, H.lib.gca_conv_halo_occupancy(tm, tn, m, lds)) for lds in (40000, 46000, 53000, 54128, 54600, 56000, 79000, 81000)])
