"""Loader/builder for ``libdsgcn.so`` — the C-ABI HIP library (declared in ``include/dsgcn.h``).

The library is built in-tree (``ds-gcn_amd/lib/libdsgcn.so``) with ``hipcc --offload-arch=gfx950`` and
loaded with ``ctypes``: no torch types cross the boundary, only raw device pointers, sizes and the
HIP stream handle.  ``lib()`` raises if the library is missing — there is no fallback.
"""
import ctypes
import glob
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB_DIR = os.path.join(_HERE, 'lib')
LIB_PATH = os.path.join(LIB_DIR, 'libdsgcn.so')
LAB_LIB_PATH = os.path.join(LIB_DIR, 'libdsgcn_lab.so')
INCLUDE = os.path.join(os.path.dirname(_HERE), 'include')

_lib = None
_lab = None

c_f = ctypes.c_void_p      # const float* / float*  (device)
c_i = ctypes.c_void_p      # const int*             (device)
c_int = ctypes.c_int
c_st = ctypes.c_void_p     # hipStream_t

# name -> argtypes; every entry point returns int (0 ok, >0 hipError_t, <0 argument error)
SIGNATURES = {
    'dsgcn_version': [],
    'dsgcn_aggregate_fwd': [c_f, c_f, c_f, c_int, c_f, c_f, c_int, c_int, c_int, c_int, c_st],
    'dsgcn_aggregate_bwd_partial_rows': [c_int, c_int, c_int],
    'dsgcn_aggregate_bwd': [c_f, c_f, c_f, c_int, c_f, c_f, c_f, c_f, c_f, c_int, c_int, c_int, c_int, c_st],
    'dsgcn_pwconv_plan': [c_int, c_int, c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                          ctypes.POINTER(ctypes.c_int)],
    'dsgcn_pwconv_fwd': [c_f] * 6 + [c_int] + [c_f] * 5 + [c_int] * 8 + [c_st],
    'dsgcn_bn_coef_rows': [c_f, c_int, c_int, c_int, c_int, c_int, c_f, c_f, c_f, ctypes.c_float, ctypes.c_double, c_int, c_f, c_int, c_st],
    'dsgcn_pwconv_fwd_ws': [c_f] * 6 + [c_int] + [c_f] * 5 + [c_int] * 8 + [c_f, c_st],
    'dsgcn_pwconv_wsplit': [c_f, c_int, c_int, c_f, c_st],
    'dsgcn_pwconv_wsplit_multi': [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int),
                                  ctypes.POINTER(ctypes.c_int), c_int, c_st],
    'dsgcn_pwconv_wsplit_bytes': [c_int] * 6,
    'dsgcn_tconv_ws_bytes': [c_int] * 7,
    'dsgcn_tconv_wsplit': [c_f, c_int, c_int, c_int, c_f, c_st],
    'dsgcn_tconv_wsplit_multi': [ctypes.c_void_p, c_int, c_st],
    'dsgcn_tconv_rows': [c_int] * 8,
    'dsgcn_tconv_fwd': [c_f] * 6 + [c_int] + [c_f] * 4 + [c_int] * 7 + [c_st],
    'dsgcn_tconv_dgrad': [c_f] * 6 + [c_int] + [c_f] * 8 + [c_int] * 7 + [c_st],
    'dsgcn_tconv_wgrad_splits': [c_int] * 7,
    'dsgcn_tconv_wgrad': [c_f] * 6 + [c_int] + [c_f] * 4 + [ctypes.c_void_p] * 2 + [c_int] * 8 + [c_st],
    'dsgcn_bn_finalize': [c_f, c_int, c_int, ctypes.c_double, c_f, c_f, ctypes.c_float, c_f, c_f, c_f, c_f, c_int,
                          c_st],
    'dsgcn_pwconv_partial_rows': [c_int] * 7,
    'dsgcn_pwconv_ipart_rows': [c_int] * 6,
    'dsgcn_dz_eff_aug': [c_f] * 7 + [c_int] * 4 + [c_st],
    'dsgcn_colsum': [c_f, c_int, c_int, c_f, c_st],
    'dsgcn_colsum_t': [c_f, c_int, c_int, c_int, c_f, c_st],
    'dsgcn_colsum_blocks': [c_f, c_int],
    'dsgcn_colsum_multi': [c_f, c_int, c_int, c_st],
    'dsgcn_colsum2': [c_f, c_int, c_int, c_int, c_f, c_f, c_int, c_int, c_int, c_f, c_st],
    'dsgcn_pwconv_dgrad': [c_f] * 6 + [c_int] + [c_f] * 10 + [c_int] * 7 + [c_st],
    'dsgcn_pwconv_dgrad_ws': [c_f] * 6 + [c_int] + [c_f] * 10 + [c_int] * 7 + [c_f, c_st],
    'dsgcn_pwconv_wgrad_splits': [c_int] * 6,
    'dsgcn_pwconv_bwd_rows': [c_int] * 6,
    'dsgcn_pwconv_bwd': [c_f] * 6 + [c_int] + [c_f] * 10 + [c_int] * 6 + [c_st],
    'dsgcn_pwconv_wgrad': [c_f] * 6 + [c_int] + [c_f] * 8 + [c_int] * 8 + [c_st],
    'dsgcn_bn_bwd_coef': [c_f] * 5 + [ctypes.c_float, ctypes.c_double, c_int, c_int] + [c_f] * 4 + [c_st],
    'dsgcn_branch_act_fwd': [c_f] * 4 + [c_int] + [c_f] + [c_int] * 4 + [c_st],
    'dsgcn_branch_act_bwd': [c_f] * 4 + [c_int] + [c_f] * 4 + [c_int] * 4 + [c_st],
    'dsgcn_tms_combine_fwd': [c_f] * 4 + [c_int] * 4 + [c_st],
    'dsgcn_tms_combine_bwd': [c_f] * 7 + [c_int] * 4 + [c_st],
    'dsgcn_tapconv_fwd': [c_f, c_f] + [c_int] * 8 + [c_i] * 6 + [ctypes.c_void_p, ctypes.c_void_p, c_st],
    'dsgcn_tapconv_dgrad': [c_f, c_f, c_f] + [c_int] * 8 + [c_i] * 6 + [ctypes.c_void_p, c_st],
    'dsgcn_tapconv_wgrad_splits': [c_int] * 8 + [c_i] * 4,
    'dsgcn_tapconv_wgrad': [c_f, c_f] + [c_int] * 8 + [c_i] * 6 + [ctypes.c_void_p, ctypes.c_void_p, c_int, c_int, c_st],
    'dsgcn_tms_rows': [c_int] * 8 + [c_i] * 3 + [c_int],
    'dsgcn_tms_fwd': [c_f] * 4 + [c_int] + [c_f] * 4 + [c_int] * 7 + [c_i] * 4 + [ctypes.c_void_p, ctypes.c_void_p, c_st],
    'dsgcn_tms_dgrad': [c_f] * 4 + [c_int] + [c_f] * 10 + [c_int] * 7 + [c_i] * 4 + [ctypes.c_void_p, c_st],
    'dsgcn_tms_wgrad': [c_f] * 4 + [c_int] + [c_f] * 5 + [c_int] * 7 + [c_i] * 4 + [ctypes.c_void_p, ctypes.c_void_p, c_int, c_st],
    'dsgcn_tms_split_rows': [c_int] * 8 + [c_i] * 4,
    'dsgcn_tms_split_fwd': [c_f] * 4 + [c_int] + [c_f] * 4 + [c_int] * 6 + [c_i] * 4 + [ctypes.c_void_p, ctypes.c_void_p, c_st],
    'dsgcn_tms_split_prep': [c_f] * 9 + [c_int] * 4 + [c_st],
    'dsgcn_tms_split_dgrad': [c_f] * 4 + [c_int] + [c_f] * 5 + [c_int] * 6 + [c_i] * 4 + [ctypes.c_void_p, c_st],
    'dsgcn_tms_split_wgrad': [c_f] * 4 + [c_int] + [c_f] * 2 + [c_int] * 6 + [c_i] * 4 + [ctypes.c_void_p, ctypes.c_void_p, c_int, c_int, c_st],
    'dsgcn_aggsum_partial_rows': [c_int, c_int, c_int],
    'dsgcn_aggsum_bwd_piece_rows': [c_int] * 5,
    'dsgcn_aggsum_fwd': [c_f, c_f] + [ctypes.c_long] * 3 + [c_f, c_f] + [c_int] * 5 + [c_st],
    'dsgcn_aggsum_bwd': [c_f, c_f] + [ctypes.c_long] * 3 + [c_f] * 6 + [ctypes.c_long] * 3 + [c_int] * 5 + [c_st],
    'dsgcn_tanhdiff_fwd': [c_f, c_f] + [c_int] * 4 + [c_st],
    'dsgcn_tanhdiff_bwd': [c_f, c_f, c_f] + [c_int] * 4 + [c_st],
    'dsgcn_tanhdiff_bwd_k': [c_f, ctypes.c_void_p, c_f] + [c_int] * 4 + [c_st],
    'dsgcn_tanhdiff_aug_fwd': [c_f, c_f, c_f] + [c_int] * 4 + [c_st],
    'dsgcn_tanhdiff_aug_bwd': [c_f, ctypes.c_void_p, c_f, c_f] + [c_int] * 4 + [c_st],
    'dsgcn_ctr_wprep': [ctypes.c_void_p, ctypes.c_void_p, c_f, c_f, c_f] + [c_int] * 3 + [c_st],
    'dsgcn_ctr_wprep_multi': [ctypes.c_void_p, c_int, c_st],
    'dsgcn_ctr_wfin_multi': [ctypes.c_void_p, c_int, c_st],
    'dsgcn_ctr_wfin': [ctypes.c_void_p, ctypes.c_void_p, c_int, ctypes.c_void_p, c_f] + [c_int] * 3 + [c_st],
    'dsgcn_ctr_affine_fwd': [ctypes.c_void_p, c_f, c_int, c_f, c_f, c_f, c_f] + [c_int] * 4 + [c_st],
    'dsgcn_ctr_affine_bwd': [ctypes.c_void_p, c_f, c_int, c_f, ctypes.c_void_p, c_f] + [c_int] * 4 + [c_st],
    'dsgcn_edge_select_fwd': [c_f, c_i, c_f] + [c_int] * 4 + [c_st],
    'dsgcn_edge_select_bwd': [c_f, c_i, c_f] + [c_int] * 4 + [c_st],
    'dsgcn_plane_stats': [c_f, c_f, ctypes.c_long, c_int, c_st],
    'dsgcn_add3': [c_f, c_f, c_f, c_f, ctypes.c_long, c_st],
    'dsgcn_pack': [c_f, c_f, c_f, c_int, c_f, c_st],
    'dsgcn_pack_fill': [c_f, c_f, c_f, c_int, c_f, c_st],
    'dsgcn_colsum_multi_host': [ctypes.c_void_p, c_int, c_st],
    'dsgcn_fuse_out_fwd': [c_f] * 6 + [c_int] + [c_f] * 2 + [c_int] * 5 + [c_st],
    'dsgcn_fuse_out_bwd': [c_f] * 6 + [c_int] + [c_f] * 5 + [c_int] * 5 + [c_st],
    'dsgcn_fuse_out_bwd3': [c_f] * 6 + [c_int] + [c_f] * 7 + [c_int] * 5 + [c_st],
    'dsgcn_fuse_out_fwd2': [c_f] * 6 + [c_int] + [c_f] * 3 + [c_int] * 5 + [c_st],
    'dsgcn_fuse_out_bwd3s': [c_f] * 6 + [c_int] + [c_f] * 3 + [c_int] + [c_f] * 4 + [c_int] * 5 + [c_st],
    'dsgcn_fuse_out_pool_fwd': [c_f] * 6 + [c_int] + [c_f] + [c_int] * 4 + [c_st],
    'dsgcn_fuse_out_pool_bwd': [c_f] * 6 + [c_int] + [c_f] * 4 + [c_int] * 4 + [c_st],
    'dsgcn_dwcausal_fwd': [c_f, c_f, c_f, c_i, c_f] + [c_int] * 6 + [c_st],
    'dsgcn_dwcausal_bwd': [c_f, c_f, c_i, c_f, c_f, c_f] + [c_int] * 6 + [c_st],
    'dsgcn_gate_fwd': [c_f, c_f, c_int, c_f, c_f] + [c_int] * 5 + [c_st],
    'dsgcn_gate_bwd': [c_f, c_f, c_int, c_f, c_f, c_int, c_f, c_f] + [c_int] * 4 + [c_st],
    'dsgcn_skeleton_prep': [ctypes.c_void_p] * 11 + [c_int] * 10 + [c_st],
    'dsgcn_dynadj_partial_stride': [c_int, c_int, c_int],
    'dsgcn_dynadj_fwd': [c_f] * 6 + [c_i, c_i, c_f] + [c_int] * 6 + [c_st],
    'dsgcn_dynadj_bwd': [c_f] * 5 + [c_i, c_i] + [c_f] * 4 + [c_int] * 7 + [c_st],
    'dsgcn_pwconv_group_ok': [c_int] * 5,
    'dsgcn_pwconv_fwd_group': [ctypes.c_void_p] * 3 + [c_int] + [ctypes.c_void_p] * 2 + [c_int] * 6 + [c_st],
    'dsgcn_pwconv_dgrad_group': [ctypes.c_void_p] * 3 + [c_int] + [ctypes.c_void_p] * 4 + [c_int] * 6 + [c_st],
    'dsgcn_pwconv_wgrad_group': [ctypes.c_void_p] * 3 + [c_int] + [ctypes.c_void_p] * 3 + [c_int] * 7 + [c_st],
    'dsgcn_pwconv_wgrad_jobs': [c_f] * 6 + [c_int] + [c_f] * 8 + [c_int] * 8 + [ctypes.c_void_p, c_int, c_st],
    'dsgcn_tms_split_wgrad_jobs': [c_f] * 4 + [c_int] + [c_f] * 2 + [c_int] * 6 + [c_i] * 4 + [ctypes.c_void_p, ctypes.c_void_p, c_int, c_int, ctypes.c_void_p, c_int, c_st],
    'dsgcn_fuse_out_fwd_drop': [c_f] * 6 + [c_int] + [c_f] * 4 + [c_int] * 5 + [ctypes.c_void_p, c_st],
    'dsgcn_fuse_out_bwd_drop': [c_f] * 6 + [c_int] + [c_f] * 3 + [c_int] + [c_f] * 4 + [c_int] * 5 + [ctypes.c_void_p, c_st],
    'dsgcn_dropout_mask': [c_f, ctypes.c_long, ctypes.c_void_p, c_st],
    'dsgcn_bn_finalize_multi': [ctypes.c_void_p, c_int, c_st],
    'dsgcn_bn_coef_rows_multi': [ctypes.c_void_p, c_int, c_st],
    'dsgcn_dynadj_fwd_jobs': [c_f] * 6 + [c_i, c_i, c_f] + [c_int] * 6 + [ctypes.c_void_p, c_int, c_st],
    'dsgcn_dynadj_bwd_jobs': [c_f] * 5 + [c_i, c_i] + [c_f] * 4 + [c_int] * 7 + [ctypes.c_void_p, c_int, c_st],
    'dsgcn_head_loss_fwd': [c_f, c_f, c_f, c_i] + [c_int] * 4 + [ctypes.c_float] + [c_f] * 6 + [c_st],
    'dsgcn_head_loss_bwd': [c_f, c_f, c_f, c_i, c_f] + [c_int] * 4 + [ctypes.c_float] + [c_f] * 3 + [c_st],
    'dsgcn_data_bn_fwd': [c_f] * 10 + [c_int] * 7 + [ctypes.c_float, ctypes.c_float, c_st],
    'dsgcn_data_bn_bwd': [c_f] * 6 + [c_int] * 6 + [c_st],
    'dsgcn_sgd_step': [c_f, c_f, c_f, c_f, ctypes.c_float, ctypes.c_float, c_int, ctypes.c_longlong, c_st],
    'dsgcn_bn_running_multi': [ctypes.c_void_p] * 5 + [ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_float),
                               ctypes.POINTER(ctypes.c_float), c_int, c_st],
}




class BnFinJob(ctypes.Structure):
    """include/dsgcn_jobs.h: dsgcn_bn_fin_job"""
    _fields_ = [('partial', ctypes.c_void_p), ('gamma', ctypes.c_void_p), ('beta', ctypes.c_void_p),
                ('mean', ctypes.c_void_p), ('var', ctypes.c_void_p), ('scale', ctypes.c_void_p), ('shift', ctypes.c_void_p),
                ('count', ctypes.c_double), ('eps', ctypes.c_float), ('nblk', ctypes.c_int), ('C', ctypes.c_int),
                ('c_affine', ctypes.c_int)]


class BnCoefJob(ctypes.Structure):
    """include/dsgcn_jobs.h: dsgcn_bn_coef_job"""
    _fields_ = [('part', ctypes.c_void_p), ('mean', ctypes.c_void_p), ('var', ctypes.c_void_p), ('gamma', ctypes.c_void_p),
                ('coef', ctypes.c_void_p), ('count', ctypes.c_double), ('eps', ctypes.c_float), ('R', ctypes.c_int),
                ('C', ctypes.c_int), ('k', ctypes.c_int), ('i_ds', ctypes.c_int), ('i_dh', ctypes.c_int),
                ('c_affine', ctypes.c_int), ('accumulate', ctypes.c_int)]


BN_JOBS_MAX = 4


class Dropout(ctypes.Structure):
    """include/dsgcn_jobs.h: dsgcn_dropout"""
    _fields_ = [('step', ctypes.c_void_p), ('seed', ctypes.c_ulonglong), ('call', ctypes.c_uint), ('p', ctypes.c_float)]


class CtrPrepJob(ctypes.Structure):
    """include/dsgcn_jobs.h: dsgcn_ctr_prep_job"""
    _fields_ = [('w', ctypes.c_void_p * 4), ('b', ctypes.c_void_p * 4), ('alpha', ctypes.c_void_p), ('wout', ctypes.c_void_p),
                ('sh', ctypes.c_void_p), ('K', ctypes.c_int), ('Co', ctypes.c_int), ('R', ctypes.c_int),
                ('reserved', ctypes.c_int)]


class CtrFinJob(ctypes.Structure):
    """include/dsgcn_jobs.h: dsgcn_ctr_fin_job"""
    _fields_ = [('dwp', ctypes.c_void_p * 4), ('ds', ctypes.c_void_p * 4), ('out', ctypes.c_void_p * 4),
                ('dalpha', ctypes.c_void_p), ('K', ctypes.c_int), ('Co', ctypes.c_int), ('R', ctypes.c_int),
                ('ds_stride', ctypes.c_int)]


class TsplitJob(ctypes.Structure):
    """include/dsgcn_jobs.h: dsgcn_tsplit_job"""
    _fields_ = [('w', ctypes.c_void_p), ('ws', ctypes.c_void_p), ('Ci', ctypes.c_int), ('Co', ctypes.c_int),
                ('KT', ctypes.c_int), ('reserved', ctypes.c_int)]


SIZE_T_RESULTS = {'dsgcn_pwconv_wsplit_bytes', 'dsgcn_tconv_ws_bytes'}      # everything else returns an int status / count

# measurement-only entry points: exported by libdsgcn_lab.so only (include/dsgcn_lab.h)
LAB_SIGNATURES = {
    'dsgcn_aggregate_fwd_valu': [c_f, c_f, c_f, c_int, c_f, c_f, c_int, c_int, c_int, c_int, c_st],
    'dsgcn_aggregate_fwd_variant': [c_f, c_f, c_f, c_int, c_f, c_f, c_int, c_int, c_int, c_int, c_int, c_st],
    'dsgcn_set_tuning': [c_int, c_int],
    'dsgcn_pwconv_tuning': [c_int, c_int],
    'dsgcn_diag_mfma_probe': [c_f, c_int, c_int, c_int, c_st],
    'dsgcn_aggsum_tuning': [c_int, c_int],
    'dsgcn_tms_tuning': [c_int, c_int],
    'dsgcn_dynadj_phases': [ctypes.c_void_p],
    'dsgcn_pwg2_phases': [ctypes.c_void_p],
    'dsgcn_pwg2_phases_block': [c_int],
    'dsgcn_bwd64_phases': [ctypes.c_void_p],
    'dsgcn_tcw_phases': [ctypes.c_void_p],
    'dsgcn_tconv_tuning': [c_int, c_int],
    'dsgcn_tms_split_tuning': [c_int, c_int],
    'dsgcn_fuse_out_tuning': [c_int, c_int],
    'dsgcn_tms_split_phases': [c_int, ctypes.c_void_p],
}


def sources(lab=False):
    """Product sources; lab=True adds csrc/lab/*.hip (measurement-only kernels: never part of libdsgcn.so)."""
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    if lab:
        srcs += sorted(glob.glob(os.path.join(CSRC, 'lab', '*.hip')))
    return srcs


def _source_hash(lab=False):
    import hashlib
    h = hashlib.sha256()
    extra = sorted(glob.glob(os.path.join(CSRC, 'lab', '*.h'))) if lab else []
    for path in (sources(lab) + sorted(glob.glob(os.path.join(CSRC, '*.h'))) + extra +
                 sorted(glob.glob(os.path.join(INCLUDE, '*.h')))):
        with open(path, 'rb') as f:
            h.update(os.path.basename(path).encode())
            h.update(f.read())
    return h.hexdigest()


def _file_hash(paths):
    import hashlib
    h = hashlib.sha256()
    for path in paths:
        with open(path, 'rb') as f:
            h.update(os.path.basename(path).encode())
            h.update(f.read())
    return h.hexdigest()


def build(force=False, verbose=False, lab=False):
    """Compile every HIP source for gfx950 into one shared library (cross-compiles without a GPU).
    lab=True builds ``libdsgcn_lab.so`` instead: the same sources with -DDSGCN_LAB, which adds the measurement-only
    entry points of include/dsgcn_lab.h (used by tools/, never by the package).
    Up-to-date check = content hash of the sources (file times do not survive the copy to a GPU box).  Each source is
    compiled to its own object (cached under lib/obj by content hash of the source + headers, compiled in parallel),
    then linked."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = sources(lab)
    lib_path = LAB_LIB_PATH if lab else LIB_PATH
    stamp = lib_path + '.srchash'
    digest = _source_hash(lab)
    if not force and os.path.exists(lib_path) and os.path.exists(stamp):
        with open(stamp) as f:
            if f.read().strip() == digest:
                return lib_path
    obj_dir = os.path.join(LIB_DIR, 'obj_lab' if lab else 'obj')
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    headers = sorted(glob.glob(os.path.join(CSRC, '*.h'))) + sorted(glob.glob(os.path.join(INCLUDE, '*.h')))
    if lab:
        headers += sorted(glob.glob(os.path.join(CSRC, 'lab', '*.h')))
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I', INCLUDE, '-I', CSRC]
    if lab:
        flags.append('-DDSGCN_LAB')
        flags += [f for f in os.environ.get('DSGCN_LAB_FLAGS', '').split() if f]      # e.g. -DKA_NT_LOADS=1 for an A/B build
    jobs, objs = [], []
    for src in srcs:
        base = os.path.splitext(os.path.basename(src))[0]
        obj = os.path.join(obj_dir, f'{base}.{_file_hash([src] + headers)[:16]}.o')
        objs.append(obj)
        if force or not os.path.exists(obj):
            for old in glob.glob(os.path.join(obj_dir, base + '.*.o')):
                os.remove(old)
            jobs.append([hipcc] + flags + ['-c', src, '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        list(pool.map(run, jobs))
    run([hipcc, '--offload-arch=gfx950', '-fPIC', '-shared', '-o', lib_path] + objs)
    with open(stamp, 'w') as f:
        f.write(digest)
    return lib_path


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'dsgcn: {LIB_PATH} is missing — build it with `python -c "import __graft_entry__ as g; g.build()"`. '
                'The HIP library is the only implementation of the hot path (no CPU/eager fallback).')
        handle = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError if the symbol is not exported
            fn.argtypes = argtypes
            fn.restype = ctypes.c_size_t if name in SIZE_T_RESULTS else ctypes.c_int
        _lib = handle
    return _lib


def lab_lib():
    """``libdsgcn_lab.so`` (product + measurement-only entry points), built on first use.  tools/ only."""
    global _lab
    if _lab is None:
        # DSGCN_LAB_LIB: an A/B run of tools/ against another build of the lab library (e.g. the previous round's kernels)
        other = os.environ.get('DSGCN_LAB_LIB')
        handle = ctypes.CDLL(other or build(lab=True))
        for name, argtypes in {**SIGNATURES, **LAB_SIGNATURES}.items():
            if other and not hasattr(handle, name):
                continue                                # (an older build under A/B: entry points added since are simply absent)
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_size_t if name in SIZE_T_RESULTS else ctypes.c_int
        _lab = handle
    return _lab


class DsgcnError(RuntimeError):
    pass


def check(code, what):
    if code != 0:
        kind = 'argument rejected' if code < 0 else 'hipError_t'
        raise DsgcnError(f'{what} failed: {kind} {code}')
