"""tmc2_amd -- Python (ctypes) host-side mirror of the MI355X-native TMC2 hot path.

The product is the C-ABI library `libtmc2hip.so` (include/tmc2hip.h).  This package only binds it for
tests, bench.py and __graft_entry__.py; it contains no algorithmic code and NO CPU fallback: every
compute call goes to the HIP library, which fails with TMC2_E_NO_DEVICE when no GPU is visible.
"""
from .lib import (Tmc2Error, Context, HostGate, Frame, SegmenterParams, Patch, load_library, library_path,
                  host_kdtree_build, host_orient_normals, host_pack_flexible, host_pack_spatial_consistency, host_global_patch_allocation,
                  encoder_global_patch_allocation, host_pack_gof_records, ctc_params, segmenter_params_check, encoder_canvas_size, metrics_display, checksum_file_write, checksum_file_read, ply_info, ply_read, ply_write,
                  point_set_checksum, host_array, SharedHostArray)
from .synth import synth_cloud, synth_gof, synth_decoded_attribute
from .gof import GofEncoder, Sharder

__all__ = ["Tmc2Error", "Context", "HostGate", "Frame", "SegmenterParams", "Patch", "load_library", "library_path",
           "host_kdtree_build", "host_orient_normals", "host_pack_flexible", "host_pack_spatial_consistency", "host_global_patch_allocation", "encoder_global_patch_allocation", "host_pack_gof_records", "ctc_params", "segmenter_params_check", "encoder_canvas_size", "metrics_display", "checksum_file_write", "checksum_file_read", "ply_info", "ply_read", "ply_write", "point_set_checksum", "host_array", "SharedHostArray", "synth_cloud", "synth_gof", "synth_decoded_attribute", "GofEncoder", "Sharder"]
