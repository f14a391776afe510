"""ctypes binding of libdlsg_hip.so (include/dlsg.h) -- the only device compute path of this package.

There is no CPU or eager-PyTorch fallback here: constructing `HipOps` raises if the library is missing or no
MI355X is visible.  Tensors are only device memory + strides; every op is a hand-written HIP kernel launched on
PyTorch's current HIP stream.

All methods take torch *views*: 2-d (rows, cols) with unit inner stride and an arbitrary row stride, or 3-d
(batch, rows, cols) for batched GEMMs (the batch stride may be 0 via .expand()).
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libdlsg_hip.so')

ABI_VERSION = 8            # include/dlsg.h DLSG_ABI_VERSION this binding was written against
GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
F_ACCUM, F_BIAS, F_TANH = 1, 2, 4
F_FORCE64, F_FORCE128, F_BF16X3, F_TILE256, F_SK, F_NOSK, F_SK_BM128, F_SK_BM256 = 256, 512, 1024, 2048, 4096, 8192, 16384, 32768
F_SK_BN128 = 262144
F_SK_NOXMAP = 131072
F_SK_GIVEAWAY = 65536      # test hook (include/dlsg.h): the split tiles are finished by their last contributor alone
MAXG = 16

c_f32p = C.c_void_p
i32, i64, u32, u64, f32 = C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_float


class GemmGroup(C.Structure):
    _fields_ = [('A', c_f32p), ('B', c_f32p), ('C', c_f32p), ('lda', i64), ('ldb', i64), ('K', i32), ('N', i32),
                ('bias', c_f32p), ('ldc', i64)]


class GemmArgs(C.Structure):
    _fields_ = [('mode', i32), ('M', i32), ('N', i32), ('ldc', i32), ('ngroups', i32), ('nbatch', i32), ('flags', i32),
                ('cu_budget', i32), ('bsa', i64), ('bsb', i64), ('bsc', i64), ('alpha', f32), ('pad2_', i32),
                ('bias', c_f32p), ('skip_if', c_f32p), ('ws', c_f32p), ('ws_bytes', i64), ('err', c_f32p), ('g', GemmGroup * MAXG)]


class RowLnArgs(C.Structure):
    _fields_ = [('x', c_f32p), ('ldx', i64), ('res', c_f32p), ('ldres', i64), ('gamma', c_f32p), ('beta', c_f32p),
                ('y', c_f32p), ('ldy', i64), ('stats', c_f32p), ('pe', c_f32p), ('pe_rows', i32), ('rows', i32),
                ('n', i32), ('pre_tanh', i32), ('post_tanh', i32), ('eps', f32), ('p1', f32), ('p2', f32),
                ('seed', u64), ('site1', u32), ('site2', u32), ('seed_ptr', c_f32p)]


class RowLnBwdArgs(C.Structure):
    _fields_ = [('f', RowLnArgs), ('dy', c_f32p), ('lddy', i64), ('dx', c_f32p), ('lddx', i64), ('accum_dx', i32),
                ('dgb_part', c_f32p), ('nblk', i32)]


class O2VArgs(C.Structure):
    _fields_ = [('y', c_f32p), ('v', c_f32p), ('g_obj', c_f32p), ('b_obj', c_f32p), ('z', c_f32p), ('ml', c_f32p),
                ('ostats', c_f32p), ('S', c_f32p), ('ws', c_f32p), ('ws_bytes', i64), ('B', i32), ('T', i32),
                ('NO', i32), ('H', i32), ('nsplit', i32), ('scale', f32), ('eps', f32)]


class O2VBwdArgs(C.Structure):
    _fields_ = [(n, c_f32p) for n in ('y', 'ostats', 'g_obj', 'b_obj', 'v', 'z', 'dz', 'S', 'ml', 'pd', 'm12', 'dy', 'dv',
                                      'part', 'ws')] + [('ws_bytes', i64), ('B', i32), ('T', i32), ('NO', i32), ('H', i32),
                                                        ('nsplit', i32), ('scale', f32), ('dysum', c_f32p)]


class LatentPslArgs(C.Structure):
    _fields_ = [(n, c_f32p) for n in ('ov', 'theta', 'gamma', 'beta', 'adj', 'u', 'out', 'stats')] + \
               [('B', i32), ('T', i32), ('P', i32), ('H', i32), ('p', f32), ('site', u32), ('eps', f32), ('pad_', f32),
                ('seed', u64), ('seed_ptr', c_f32p)]


class SaCoreArgs(C.Structure):
    _fields_ = [(n, c_f32p) for n in ('K', 'Q', 'V', 'mask', 'w', 'out')] + [('B', i32), ('T', i32), ('D', i32), ('scale', f32)]


class BeamSelectArgs(C.Structure):
    _fields_ = [('logits', c_f32p), ('ld', i64), ('last', c_f32p), ('last_lp', c_f32p), ('pred', c_f32p), ('new_lp', c_f32p),
                ('back', c_f32p), ('rows', c_f32p), ('ended_count', c_f32p), ('B', i32), ('k', i32), ('V', i32), ('end', i32),
                ('first', i32), ('pad_', i32)]


class GatherMultiArgs(C.Structure):
    _fields_ = [('src', C.c_void_p * 4), ('dst', C.c_void_p * 4), ('n', i32 * 4), ('rows', c_f32p), ('nrows', i32),
                ('count', i32)]


class SaCoreBwdArgs(C.Structure):
    _fields_ = [(n, c_f32p) for n in ('w', 'K', 'Q', 'V', 'dout', 'dK', 'dQ', 'dV')] + [('B', i32), ('T', i32), ('D', i32),
                                                                                          ('scale', f32)]


class LatentPslBwdArgs(C.Structure):
    _fields_ = [(n, c_f32p) for n in ('dout', 'u', 'stats', 'gamma', 'adj', 'ov', 'theta', 'dov', 'dtheta_part', 'part')] + \
               [('B', i32), ('T', i32), ('P', i32), ('H', i32), ('p', f32), ('site', u32), ('seed', u64), ('seed_ptr', c_f32p)]


class BilstmArgs(C.Structure):
    _fields_ = [('xg', C.c_void_p * 2), ('ldxg', i64), ('w_hh', C.c_void_p * 2), ('b_ih', C.c_void_p * 2), ('b_hh', C.c_void_p * 2),
                ('out', c_f32p), ('hprev', C.c_void_p * 2), ('c', C.c_void_p * 2), ('gates', C.c_void_p * 2), ('hx', c_f32p),
                ('flags', c_f32p), ('err', c_f32p), ('B', i32), ('T', i32), ('H', i32), ('pad_', i32)]


class BilstmBwdArgs(C.Structure):
    _fields_ = [('gates', C.c_void_p * 2), ('c', C.c_void_p * 2), ('dout', c_f32p), ('w_hh', C.c_void_p * 2),
                ('dgates', C.c_void_p * 2), ('gx', c_f32p), ('px', c_f32p), ('flags', c_f32p), ('err', c_f32p),
                ('B', i32), ('T', i32), ('H', i32), ('pad_', i32)]


class LstmSeqArgs(C.Structure):
    _fields_ = [(n, c_f32p) for n in ('addend', 'addend_out', 'W', 'As', 'Hs', 'Cs', 'dHs', 'dAs', 'dCs', 'DA', 'DH', 'DC',
                                      'gA', 'gC', 'gDH', 'gDC', 'xbuf', 'xbuf2', 'flags', 'err')] + \
               [('L', i32), ('n', i32), ('H', i32), ('batch_major', i32)] + \
               [(n, c_f32p) for n in ('b_ih', 'b_hh', 'Hprev', 'gDHprev')]


_P4 = c_f32p * 4
_P2 = c_f32p * 2


class ClnArgs(C.Structure):
    _fields_ = [('x', _P4), ('gamma', _P4), ('beta', _P4), ('y', _P4), ('dy', _P4 * 3), ('dx', _P4), ('dgamma', _P4), ('dbeta', _P4),
                ('extra', _P4), ('U', _P4), ('gx', _P4), ('gdy', _P4), ('gpart', _P4), ('ws', c_f32p),
                ('rows', i32), ('N', i32), ('groups', i32), ('pre_tanh', i32), ('ndy', i32), ('acc_lo', i32), ('acc_hi', i32), ('defer', i32),
                ('eps', f32), ('p_pre', f32), ('p_post', f32), ('site_pre', u32), ('site_post', u32), ('pad2_', u32),
                ('seed', u64), ('seed_ptr', C.c_void_p), ('row0', i64)]


class CritSaArgs(C.Structure):
    _fields_ = [(n, c_f32p) for n in ('KQV', 'smask', 'w', 'ctx', 'dctx', 'dKQV', 'U', 'Uctx', 'gKQV')] + \
               [('n', i32), ('B', i32), ('L', i32), ('acc_lo', i32), ('acc_hi', i32), ('pad_', i32), ('scale', f32), ('pad2_', f32)]


class CritPattnArgs(C.Structure):
    _fields_ = [('a', _P2), ('e', _P2), ('smask', c_f32p), ('P', _P2), ('wgt', _P2), ('aggpre', _P2), ('d_agg', _P2), ('d_wgt', _P2),
                ('da', _P2), ('de', _P2), ('Ua', _P2), ('Uagg', _P2), ('Uwgt', _P2), ('ga', _P2), ('ge', _P2),
                ('n', i32), ('B', i32), ('L', i32), ('T', i32), ('acc_lo', i32), ('acc_hi', i32), ('scale', f32), ('pad_', f32)]


class CritTsumArgs(C.Structure):
    _fields_ = [(n, c_f32p) for n in ('words', 'theta', 'gamma', 'beta', 'fusion', 'adj', 'u', 'sent', 'fus', 'd_fus', 'dwords', 'part',
                                      'U', 'Ufus', 'gwords', 'gpart')] + \
               [('n', i32), ('L', i32), ('acc_lo', i32), ('acc_hi', i32), ('eps', f32), ('p', f32), ('site', u32), ('pad_', u32),
                ('seed', u64), ('seed_ptr', C.c_void_p), ('row0', i64)]


class CritScoreArgs(C.Structure):
    _fields_ = [('v', _P2), ('s', _P2), ('wc', _P2), ('bc', _P2), ('wgt', _P2), ('fus', c_f32p), ('pair', _P2), ('score', _P2),
                ('both', c_f32p), ('out', c_f32p), ('d_out', c_f32p), ('d_fus', c_f32p), ('c_spre', _P2), ('c_vpre', _P2), ('d_wgt', _P2),
                ('part_wc', _P2), ('dbc', c_f32p), ('Uspre', _P2), ('Uwgt', _P2), ('Ufus', c_f32p), ('scratch', c_f32p),
                ('n', i32), ('B', i32), ('T', i32), ('ng', i32), ('acc_lo', i32), ('acc_hi', i32)]


class CritReduceDesc(C.Structure):
    _fields_ = [('src', c_f32p), ('stride', i64), ('n', i64), ('nslab', i32), ('scale', f32), ('out', c_f32p)]


class CritColsumDesc(C.Structure):
    _fields_ = [('part', c_f32p), ('ld', i64), ('rows', i32), ('n', i32), ('part_b', c_f32p), ('ld_b', i64), ('rows_b', i32), ('pad_', i32),
                ('out', c_f32p), ('out_b', c_f32p), ('scale', f32), ('pad2_', f32)]


class ColsumDesc(C.Structure):
    _fields_ = [('part', c_f32p), ('ld', i64), ('out_a', c_f32p), ('out_b', c_f32p), ('rows', i32), ('n', i32), ('split', i32),
                ('dup', i32), ('accum', i32), ('pad_', i32)]


class DecAttArgs(C.Structure):
    _fields_ = [('Kp', c_f32p * 2), ('Vp', c_f32p * 2), ('q', c_f32p), ('ldq', i64), ('c', c_f32p * 2), ('ldc', i64),
                ('alpha', c_f32p), ('B', i32), ('P', i32), ('Q', i32), ('H', i32), ('nstream', i32), ('scale', f32)]


class DecAttBwdArgs(C.Structure):
    _fields_ = [('f', DecAttArgs), ('dc', c_f32p * 2), ('lddc', i64), ('dalpha', c_f32p), ('dKp', c_f32p * 2),
                ('dVp', c_f32p * 2), ('dq', c_f32p), ('lddq', i64), ('accum_dq', i32)]


class LstmPwArgs(C.Structure):
    _fields_ = [('slabs', c_f32p), ('nslab', i32), ('pad_', i32), ('slab_stride', i64), ('addend', c_f32p),
                ('ldadd', i64), ('b_ih', c_f32p), ('b_hh', c_f32p), ('c_prev', c_f32p), ('ldcp', i64), ('c', c_f32p),
                ('ldc_', i64), ('h', c_f32p), ('ldh', i64), ('h2', c_f32p), ('ldh2', i64), ('gates', c_f32p), ('ldg', i64),
                ('B', i32), ('H', i32), ('p', f32), ('site', u32), ('seed', u64), ('seed_ptr', c_f32p)]


class LstmPwBwdArgs(C.Structure):
    _fields_ = [('gates', c_f32p), ('ldg', i64), ('c', c_f32p), ('ldc_', i64), ('c_prev', c_f32p), ('ldcp', i64), ('dh', c_f32p),
                ('lddh', i64), ('dh2', c_f32p), ('lddh2', i64), ('dh3', c_f32p), ('lddh3', i64), ('dh4', c_f32p), ('lddh4', i64),
                ('dh4_nslab', i32), ('pad4_', i32), ('dh4_slab_stride', i64),
                ('dc_next', c_f32p), ('lddcn', i64),
                ('dgates', c_f32p), ('lddg', i64), ('dc_prev', c_f32p), ('lddcp', i64), ('B', i32), ('H', i32), ('p', f32),
                ('site', u32), ('seed', u64), ('seed_ptr', c_f32p)]


class DecMidArgs(C.Structure):
    _fields_ = [('slabs', c_f32p), ('nslab', i32), ('pad_', i32), ('slab_stride', i64), ('addend', c_f32p), ('ldadd', i64),
                ('b_ih', c_f32p), ('b_hh', c_f32p), ('c_prev', c_f32p), ('c', c_f32p), ('h', c_f32p), ('gates', c_f32p),
                ('lnq_g', c_f32p), ('lnq_b', c_f32p), ('qcur', c_f32p), ('st_q', c_f32p), ('p_q', f32), ('site_q', u32),
                ('Kp', C.c_void_p * 2), ('Vp', C.c_void_p * 2), ('lnc_g', C.c_void_p * 2), ('lnc_b', C.c_void_p * 2),
                ('cpre', C.c_void_p * 2), ('ctx', C.c_void_p * 2), ('st_c', C.c_void_p * 2), ('alpha', c_f32p),
                ('p_att', f32 * 2), ('site_att', u32 * 2), ('B', i32), ('Q', i32), ('H', i32), ('P', i32), ('nstream', i32),
                ('scale', f32), ('eps', f32), ('kv_div', i32), ('seed', u64), ('seed_ptr', c_f32p)]


class DecTailArgs(C.Structure):
    _fields_ = [('slabs', c_f32p), ('nslab', i32), ('pad_', i32), ('slab_stride', i64), ('b_ih', c_f32p), ('b_hh', c_f32p),
                ('c_prev', c_f32p), ('c', c_f32p), ('hd', c_f32p), ('gates', c_f32p), ('ln_g', c_f32p), ('ln_b', c_f32p),
                ('dout', c_f32p), ('st_l', c_f32p), ('p', f32), ('site', u32), ('B', i32), ('D', i32), ('eps', f32),
                ('seed', u64), ('seed_ptr', c_f32p),
                ('s_coins', c_f32p), ('s_t', i32), ('s_V', i32), ('s_W', c_f32p), ('s_b', c_f32p), ('s_E', c_f32p), ('s_Wd', i32),
                ('s_site', u32), ('s_ids', c_f32p), ('s_we', c_f32p), ('s_ldwe', i64), ('s_row0', i64), ('s_p', f32), ('pad2_', f32)]


class DecMidBwdArgs(C.Structure):
    _fields_ = [('slabs', c_f32p), ('nslab', i32), ('write_rec', i32), ('slab_stride', i64), ('dlh_rec', c_f32p),
                ('cpre', C.c_void_p * 2), ('st_c', C.c_void_p * 2), ('lnc_g', C.c_void_p * 2), ('part_c', C.c_void_p * 2),
                ('dcpre', C.c_void_p * 2), ('p_att', f32 * 2), ('site_att', u32 * 2), ('Kp', C.c_void_p * 2),
                ('Vp', C.c_void_p * 2), ('alpha', c_f32p), ('dalpha', c_f32p), ('ds', c_f32p), ('qh', c_f32p),
                ('st_q', c_f32p), ('lnq_g', c_f32p), ('part_q', c_f32p), ('p_q', f32), ('site_q', u32),
                ('rec_slabs', c_f32p), ('rec_nslab', i32), ('pad_', i32), ('rec_slab_stride', i64), ('rec_ld', i64),
                ('gates', c_f32p), ('c', c_f32p), ('c_prev', c_f32p), ('dc', c_f32p), ('dgates', c_f32p),
                ('B', i32), ('Q', i32), ('H', i32), ('D', i32), ('P', i32), ('nstream', i32), ('scale', f32), ('pad2_', f32),
                ('seed', u64), ('seed_ptr', c_f32p)]


class DecattCacheGradsArgs(C.Structure):
    _fields_ = [('alpha', c_f32p), ('ds', c_f32p), ('qcur', c_f32p), ('dcpre', C.c_void_p * 2), ('dKp', C.c_void_p * 2),
                ('dVp', C.c_void_p * 2), ('L', i32), ('B', i32), ('Q', i32), ('H', i32), ('P', i32), ('nstream', i32)]


# every symbol include/dlsg.h declares (checked by tests/test_abi.py against the header text)
SYMBOLS = ['dlsg_abi_version', 'dlsg_struct_size', 'dlsg_gemm', 'dlsg_gemm_variant', 'dlsg_gemm_ws_bytes', 'dlsg_slab_reduce', 'dlsg_rowln_fwd', 'dlsg_rowln_bwd', 'dlsg_rowln_fwd_multi', 'dlsg_rowln_bwd_multi',
           'dlsg_rowln_bwd_nblk', 'dlsg_colsum', 'dlsg_colsum2', 'dlsg_colsum_ws_floats', 'dlsg_colsum_multi', 'dlsg_colsum_multi_ok', 'dlsg_o2v_workspace_bytes', 'dlsg_o2v_fwd', 'dlsg_o2v_fwd_multi',
           'dlsg_softmax_fwd', 'dlsg_softmax_bwd', 'dlsg_decatt_fwd', 'dlsg_decatt_bwd', 'dlsg_lstm_pw_fwd',
           'dlsg_lstm_pw_bwd', 'dlsg_lstm_pw_fwd_n', 'dlsg_lstm_pw_bwd_n', 'dlsg_mean_rows_fwd', 'dlsg_mean_rows_bwd', 'dlsg_embed_fwd', 'dlsg_embed_bwd',
           'dlsg_argmax', 'dlsg_select_embed', 'dlsg_copy2d', 'dlsg_dropout', 'dlsg_fill', 'dlsg_ce_ragged', 'dlsg_log_softmax',
           'dlsg_adam', 'dlsg_permute_tb', 'dlsg_gather_rows', 'dlsg_dec_mid_fwd', 'dlsg_dec_tail_fwd',
           'dlsg_dec_mid_bwd', 'dlsg_decatt_cache_grads', 'dlsg_o2v_bwd', 'dlsg_o2v_bwd_multi',
           'dlsg_latent_psl_fwd', 'dlsg_sa_core_fwd', 'dlsg_beam_select', 'dlsg_gather_rows_multi',
           'dlsg_sa_core_bwd', 'dlsg_latent_psl_bwd', 'dlsg_latent_psl_fwd_multi', 'dlsg_latent_psl_bwd_multi',
           'dlsg_crit_embed_mix', 'dlsg_crit_embed_mix_bwd', 'dlsg_crit_vocab_scatter', 'dlsg_crit_relu_taps', 'dlsg_crit_relu_taps_bwd',
           'dlsg_cln_ws_floats', 'dlsg_cln_fwd', 'dlsg_cln_bwd', 'dlsg_cln_bwd2', 'dlsg_crit_sa_fwd', 'dlsg_crit_sa_bwd', 'dlsg_crit_sa_bwd2',
           'dlsg_crit_pattn_fwd', 'dlsg_crit_pattn_bwd', 'dlsg_crit_pattn_bwd2', 'dlsg_crit_tsum_fwd', 'dlsg_crit_tsum_bwd',
           'dlsg_crit_tsum_bwd2', 'dlsg_crit_score_fwd', 'dlsg_crit_score_bwd', 'dlsg_crit_score_bwd2', 'dlsg_crit_gp', 'dlsg_crit_topk',
           'dlsg_crit_unselect', 'dlsg_crit_colsum', 'dlsg_crit_reduce',
           'dlsg_bilstm_supported', 'dlsg_bilstm_hx_floats', 'dlsg_bilstm_flag_words', 'dlsg_bilstm_fwd', 'dlsg_bilstm_bwd_x_floats', 'dlsg_bilstm_bwd',
           'dlsg_lstm_seq_supported', 'dlsg_lstm_seq_x_floats', 'dlsg_lstm_seq_flag_words', 'dlsg_lstm_seq',
           'dlsg_comm_unique_id', 'dlsg_comm_init', 'dlsg_comm_destroy', 'dlsg_comm_info', 'dlsg_allreduce_bucket',
           'dlsg_allreduce_buckets', 'dlsg_allreduce_max_i32', 'dlsg_comm_rehearsal', 'dlsg_comm_async_error']


def load_library(path=LIB_PATH):
    if not os.path.exists(path):
        raise RuntimeError('libdlsg_hip.so not built (%s): run `python -c "import __graft_entry__ as g; g.build()"` '
                           'or `make -C d-lsg-video-caption_amd/csrc`; there is no fallback path' % path)
    lib = C.CDLL(path)
    for s in SYMBOLS:
        getattr(lib, s)  # AttributeError if the library does not export what the header declares
    vp, P = C.c_void_p, C.POINTER
    sig = {
        'dlsg_abi_version': [],
        'dlsg_struct_size': [i32],
        'dlsg_gemm': [P(GemmArgs), vp],
        'dlsg_gemm_variant': [P(GemmArgs)],
        'dlsg_gemm_ws_bytes': [],
        'dlsg_slab_reduce': [vp, i32, i64, vp, vp, i64, i32, i32, i32, vp],
        'dlsg_rowln_fwd': [P(RowLnArgs), vp],
        'dlsg_rowln_bwd': [P(RowLnBwdArgs), vp],
        'dlsg_rowln_fwd_multi': [P(RowLnArgs), i32, vp],
        'dlsg_rowln_bwd_multi': [P(RowLnBwdArgs), i32, vp],
        'dlsg_rowln_bwd_nblk': [i32],
        'dlsg_colsum_ws_floats': [i32, i32],
        'dlsg_colsum': [vp, i64, i32, i32, vp, i32, vp, vp],
        'dlsg_colsum2': [vp, i64, i32, i32, vp, vp, i32, i32, i32, vp, vp],
        'dlsg_colsum_multi': [P(ColsumDesc), i32, vp],
        'dlsg_colsum_multi_ok': [vp, i64, i32, i32],
        'dlsg_o2v_workspace_bytes': [i32, i32, i32, i32],
        'dlsg_o2v_fwd': [P(O2VArgs), vp],
        'dlsg_o2v_fwd_multi': [P(O2VArgs), i32, vp],
        'dlsg_softmax_fwd': [vp, vp, vp, i64, i32, i32, vp],
        'dlsg_softmax_bwd': [vp, vp, vp, i64, i32, i32, vp],
        'dlsg_decatt_fwd': [P(DecAttArgs), vp],
        'dlsg_decatt_bwd': [P(DecAttBwdArgs), vp],
        'dlsg_lstm_pw_fwd': [P(LstmPwArgs), vp],
        'dlsg_lstm_pw_bwd': [P(LstmPwBwdArgs), vp],
        'dlsg_lstm_pw_fwd_n': [P(LstmPwArgs), i32, vp],
        'dlsg_lstm_pw_bwd_n': [P(LstmPwBwdArgs), i32, vp],
        'dlsg_mean_rows_fwd': [vp, vp, i64, i32, i32, i32, vp],
        'dlsg_mean_rows_bwd': [vp, i64, vp, i32, i32, i32, i32, vp],
        'dlsg_embed_fwd': [vp, vp, vp, i64, i32, i32, f32, u64, u32, i64, vp, vp],
        'dlsg_embed_bwd': [vp, i64, vp, vp, i32, i32, f32, u64, u32, i64, vp, vp],
        'dlsg_select_embed': [vp, i64, i32, vp, i32, i32, vp, vp, vp, vp, i64, i32, i32, f32, u64, u32, i64, vp, i32, vp],
        'dlsg_argmax': [vp, i64, vp, i32, i32, vp],
        'dlsg_copy2d': [vp, i64, vp, i64, i32, i32, i32, vp],
        'dlsg_dropout': [vp, i64, vp, i64, i32, i32, f32, u64, u32, vp, vp],
        'dlsg_fill': [vp, i64, f32, vp],
        'dlsg_ce_ragged': [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
        'dlsg_log_softmax': [vp, vp, i32, i32, vp],
        'dlsg_adam': [vp, vp, vp, vp, i64, f32, f32, f32, f32, i32, f32, vp, vp, vp],
        'dlsg_permute_tb': [vp, vp, i32, i32, i32, vp],
        'dlsg_gather_rows': [vp, i64, vp, vp, i64, i32, i32, vp],
        'dlsg_dec_mid_fwd': [P(DecMidArgs), vp],
        'dlsg_dec_tail_fwd': [P(DecTailArgs), vp],
        'dlsg_dec_mid_bwd': [P(DecMidBwdArgs), vp],
        'dlsg_decatt_cache_grads': [P(DecattCacheGradsArgs), vp],
        'dlsg_o2v_bwd': [P(O2VBwdArgs), vp],
        'dlsg_o2v_bwd_multi': [P(O2VBwdArgs), i32, vp],
        'dlsg_latent_psl_fwd': [P(LatentPslArgs), vp],
        'dlsg_latent_psl_fwd_multi': [P(LatentPslArgs), i32, vp],
        'dlsg_sa_core_fwd': [P(SaCoreArgs), vp],
        'dlsg_beam_select': [P(BeamSelectArgs), vp],
        'dlsg_gather_rows_multi': [P(GatherMultiArgs), vp],
        'dlsg_sa_core_bwd': [P(SaCoreBwdArgs), vp],
        'dlsg_latent_psl_bwd': [P(LatentPslBwdArgs), vp],
        'dlsg_latent_psl_bwd_multi': [P(LatentPslBwdArgs), i32, vp],
        'dlsg_bilstm_supported': [i32, i32, i32],
        'dlsg_bilstm_hx_floats': [i32, i32],
        'dlsg_bilstm_flag_words': [i32, i32],
        'dlsg_bilstm_fwd': [P(BilstmArgs), vp],
        'dlsg_bilstm_bwd_x_floats': [i32, i32],
        'dlsg_bilstm_bwd': [P(BilstmBwdArgs), vp],
        'dlsg_comm_unique_id': [vp],
        'dlsg_lstm_seq_supported': [i32, i32, i32],
        'dlsg_lstm_seq_x_floats': [i32, i32, i32],
        'dlsg_lstm_seq_flag_words': [i32, i32, i32],
        'dlsg_lstm_seq': [P(LstmSeqArgs), i32, vp],
        'dlsg_crit_embed_mix': [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
        'dlsg_crit_embed_mix_bwd': [vp, vp, vp, vp, i32, i32, i32, vp],
        'dlsg_crit_vocab_scatter': [vp, vp, vp, i32, i32, vp],
        'dlsg_crit_relu_taps': [vp, vp, vp, f32, vp, vp, i32, i32, vp],
        'dlsg_crit_relu_taps_bwd': [vp, vp, vp, vp, i32, i32, vp],
        'dlsg_cln_ws_floats': [i32, i32],
        'dlsg_cln_fwd': [P(ClnArgs), vp], 'dlsg_cln_bwd': [P(ClnArgs), vp], 'dlsg_cln_bwd2': [P(ClnArgs), vp],
        'dlsg_crit_sa_fwd': [P(CritSaArgs), vp], 'dlsg_crit_sa_bwd': [P(CritSaArgs), vp], 'dlsg_crit_sa_bwd2': [P(CritSaArgs), vp],
        'dlsg_crit_pattn_fwd': [P(CritPattnArgs), vp], 'dlsg_crit_pattn_bwd': [P(CritPattnArgs), vp],
        'dlsg_crit_pattn_bwd2': [P(CritPattnArgs), vp],
        'dlsg_crit_tsum_fwd': [P(CritTsumArgs), vp], 'dlsg_crit_tsum_bwd': [P(CritTsumArgs), vp], 'dlsg_crit_tsum_bwd2': [P(CritTsumArgs), vp],
        'dlsg_crit_score_fwd': [P(CritScoreArgs), vp], 'dlsg_crit_score_bwd': [P(CritScoreArgs), vp],
        'dlsg_crit_score_bwd2': [P(CritScoreArgs), vp],
        'dlsg_crit_gp': [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp],
        'dlsg_crit_topk': [vp, i64, i64, i32, vp, vp, i32, i32, i32, i32, vp],
        'dlsg_crit_unselect': [vp, vp, vp, i32, i32, i32, i32, vp],
        'dlsg_crit_colsum': [P(CritColsumDesc), i32, vp],
        'dlsg_crit_reduce': [P(CritReduceDesc), i32, vp],
        'dlsg_comm_init': [P(vp), vp, i32, i32],
        'dlsg_comm_destroy': [vp],
        'dlsg_comm_info': [vp, P(i32), P(i32), P(i32)],
        'dlsg_allreduce_bucket': [vp, vp, i64, vp],
        'dlsg_allreduce_buckets': [vp, P(vp), P(i64), i32, vp],
        'dlsg_allreduce_max_i32': [vp, vp, i64, vp],
        'dlsg_comm_async_error': [vp, P(i32)],
        'dlsg_comm_rehearsal': [vp, i64, i32, i32, vp],
    }
    assert sorted(sig) == sorted(SYMBOLS)
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int64 if name in ('dlsg_gemm_ws_bytes', 'dlsg_o2v_workspace_bytes', 'dlsg_cln_ws_floats', 'dlsg_colsum_ws_floats',
                                            'dlsg_bilstm_hx_floats', 'dlsg_bilstm_flag_words', 'dlsg_bilstm_bwd_x_floats',
                                            'dlsg_lstm_seq_x_floats', 'dlsg_lstm_seq_flag_words') else C.c_int
    return lib


STRUCTS = [GemmArgs, RowLnArgs, RowLnBwdArgs, O2VArgs, DecAttArgs, DecAttBwdArgs, LstmPwArgs, LstmPwBwdArgs, DecMidArgs,
           DecTailArgs, DecMidBwdArgs, DecattCacheGradsArgs, O2VBwdArgs, LatentPslArgs,
           SaCoreArgs, BeamSelectArgs, GatherMultiArgs, SaCoreBwdArgs,
           LatentPslBwdArgs, BilstmArgs, BilstmBwdArgs, ColsumDesc, LstmSeqArgs, ClnArgs, CritSaArgs, CritPattnArgs, CritTsumArgs,
           CritScoreArgs, CritColsumDesc, CritReduceDesc]


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chkc(t):
    if not t.is_contiguous():
        raise ValueError('fused decoder-step kernels take dense row-major buffers')


def _seed(seed):
    """seed is an int, or a 1-element int64 device tensor (graph mode: the kernels read it at run time)."""
    if torch.is_tensor(seed):
        return 0, C.c_void_p(seed.data_ptr())
    return int(seed), None


def host_to_device(values, dtype, device):
    """A small host array (per-step scalars: coins, seeds, Adam's bias corrections, caption lengths, gather indices) as a device
    tensor WITHOUT stalling the host.  A copy from pageable memory is synchronous on ROCm -- it waits until the stream has run dry --
    so a loop that sends three scalars per step never gets ahead of the device, and every microsecond of host work between two
    steps is device idle time (bench.py `sustained`: 0.8 ms per step).  Staged through the caching pinned allocator the copy is
    enqueued and the host moves on; the allocator keeps the block until the copy's event has passed."""
    t = values.to(dtype) if torch.is_tensor(values) else torch.as_tensor(values, dtype=dtype)
    device = torch.device(device)
    if device.type != 'cuda' or t.is_cuda:
        return t.to(device)
    return t.contiguous().pin_memory().to(device, non_blocking=True)


def copy_to_device(dst, values, dtype=None):
    """dst (device tensor) <- values, same staging as `host_to_device`"""
    t = values.to(dtype or dst.dtype) if torch.is_tensor(values) else torch.as_tensor(values, dtype=dtype or dst.dtype)
    if dst.is_cuda and not t.is_cuda:
        t = t.contiguous().pin_memory()
    dst.copy_(t.view(dst.shape), non_blocking=True)


def _chk2(t):
    assert t.dim() == 2 and (t.stride(1) == 1 or t.size(1) == 1) and t.dtype == torch.float32, (t.shape, t.stride())
    return t


class HipOps(object):
    """Launches the HIP kernels on torch's current stream.  Raises if the library or the GPU is missing."""

    name = 'hip'

    def __init__(self):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise RuntimeError('dlsg_amd needs an MI355X (gfx950) device: torch.cuda.is_available() is False and '
                               'there is no CPU fallback')
        if self.lib.dlsg_abi_version() != ABI_VERSION:
            raise RuntimeError('libdlsg_hip.so ABI mismatch: library %d, binding %d (rebuild: make -C d-lsg-video-caption_amd/csrc)'
                               % (self.lib.dlsg_abi_version(), ABI_VERSION))
        self.prof = None          # set to {} by bench.py: key -> list of (start event, end event, algorithmic work)
        self.extra_flags = 0      # OR-ed into every dlsg_gemm call (precision policy: F_BF16X3), set by the model
        self.prof_min_flops = 2e9  # dlsg_gemm calls below this are not bracketed by profile events (tools/pmc_step_target.py: 0)
        self.flop_count = None     # a float: every dlsg_gemm call adds its 2 M N K (bench.py: executed flops of a step)

    # ------------------------------------------------------------------ live per-kernel timing (bench.py roofline)
    def _prof_begin(self):
        if self.prof is None:
            return None
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        return e0

    def _prof_end(self, key, e0, work, shape='', operand_bytes=None):
        if e0 is None:
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.prof.setdefault(key, []).append((e0, e1, work, shape, operand_bytes))

    def prof_summary(self):
        """key -> dict(launches, ms_total, work_total, shapes: launch shape -> the same three); call after
        torch.cuda.synchronize().  A kernel symbol runs several launch shapes in a step; traffic counters are per shape."""
        out = {}
        for key, rec in (self.prof or {}).items():
            shapes, samples = {}, {}
            for a, b, w, shp, ob in rec:
                d = shapes.setdefault(shp, dict(launches=0, ms_total=0.0, work_total=0.0, operand_bytes=ob))
                d['launches'] += 1; d['work_total'] += float(w)
                samples.setdefault(shp, []).append(a.elapsed_time(b))
            # a shape's time = launches x the MEDIAN of its samples: the span between two events also holds whatever the host
            # took to issue the launch while the device sat idle (the first launch of an eager step: one sample of 10 ms among
            # three turned a 0.2-ms launch into "3.6 ms" in one run)
            for shp, ms in samples.items():
                ms = sorted(ms)
                med = ms[len(ms) // 2] if len(ms) % 2 else 0.5 * (ms[len(ms) // 2 - 1] + ms[len(ms) // 2])
                shapes[shp]['ms_total'] = med * len(ms)
            out[key] = dict(launches=len(rec), ms_total=sum(d['ms_total'] for d in shapes.values()),
                            work_total=float(sum(r[2] for r in rec)), shapes=shapes)
        return out

    # ------------------------------------------------------------------ plumbing
    @staticmethod
    def _stream():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    launches = 0              # entry-point calls made through this binding (bench.py: launches per decode step)

    def _check(self, rc, what):
        self.launches += 1
        if rc != 0:
            raise RuntimeError('%s failed with code %d' % (what, rc))

    # ------------------------------------------------------------------ GEMM
    stream_k = True           # False: dlsg_gemm gets no workspace, so every product runs on the tiled kernels
    sk_cu_budget = 0          # > 0: stream-K launches use at most this many workgroups (dlsg_gemm_args.cu_budget): the backward of a
                              # Trainer with several ranks leaves CUs to the bucket all-reduces that run beside it

    # process-wide (class attributes): launches on one stream are serialised, so every model, trainer and captured graph that
    # launches on a stream can share that stream's workspace; captures all run on ONE side stream per device (capture_stream), so a
    # process holds two workspaces per device (~134 MB each on 256 CUs), not one per captured graph
    _gemm_ws = {}
    _capture_streams = {}

    @classmethod
    def capture_stream(cls, dev):
        """the side stream every hipGraph capture of this process runs on (Trainer, GanTrainer, GreedyGraph, BeamGraph): captures
        are sequential and replays go to the caller's current stream, so one stream -- and one stream-K workspace, warmed by the
        eager pass each capture site runs first -- serves them all"""
        dev = torch.device(dev)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        st = cls._capture_streams.get(idx)
        if st is None:
            st = cls._capture_streams[idx] = torch.cuda.Stream(device=idx)
        return st

    def _gemm_workspace(self, dev):
        """the stream-K kernel's scratch (csrc/gemm_sk.hip): counters + two accumulator slots per CU, zero-filled once; one per
        (device, stream), because two launches that may overlap must not share it.  None while a capture is running on a stream
        that has no workspace yet (no allocation + fill inside a capture): the call then runs on the tiled kernels."""
        key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
        ws = HipOps._gemm_ws.get(key)
        if ws is None:
            n = int(self.lib.dlsg_gemm_ws_bytes())
            if n <= 0:
                raise RuntimeError('dlsg_gemm_ws_bytes() failed: no device?')
            if torch.cuda.is_current_stream_capturing():
                return None
            ws = HipOps._gemm_ws[key] = torch.zeros((n + 3) // 4, dtype=torch.float32, device=dev)
        return ws

    def device_cus(self):
        """compute units of the current device (what a stream-K launch fills: the workspace is two slots per CU)"""
        n = int(self.lib.dlsg_gemm_ws_bytes())
        return max(0, (n - 16384) // (2 * 256 * 256 * 4))

    def comm_rehearsal(self, view, workgroups, passes):
        """one-GPU stand-in for the footprint of a ring all-reduce over `view` (dlsg_comm_rehearsal), on the current stream"""
        self._check(self.lib.dlsg_comm_rehearsal(_p(view), i64(view.numel()), int(workgroups), int(passes), self._stream()),
                    'dlsg_comm_rehearsal')

    def gemm(self, mode, groups, alpha=1.0, flags=0, bias=None, skip_if=None, plan_only=False):
        """groups: list of (A, B, C[, bias]) views (2-d, or 3-d batched with identical batch strides across groups).
        plan_only: nothing is launched; returns the tile family the library would run the call on (dlsg_gemm_variant)."""
        a = GemmArgs()
        A0, B0, C0 = groups[0][:3]
        batched = A0.dim() == 3
        if batched:
            nb = C0.size(0)
            a.bsa, a.bsb, a.bsc = A0.stride(0), B0.stride(0), C0.stride(0)
        else:
            nb = 1
            a.bsa = a.bsb = a.bsc = 0
        M = C0.shape[-2]
        N = max(g[2].shape[-1] for g in groups)          # groups may write column blocks of different widths
        a.mode, a.M, a.N, a.ldc = mode, M, N, C0.stride(-2)
        a.ngroups, a.nbatch, a.alpha = len(groups), nb, alpha
        gbias = any(len(g) > 3 and g[3] is not None for g in groups)
        a.flags = flags | self.extra_flags | (F_BIAS if (bias is not None or gbias) else 0)
        a.bias = _p(bias)
        a.skip_if = _p(skip_if)          # 1-element int32 device tensor: launch is a no-op when it is non-zero
        if (self.stream_k or (flags & F_SK)) and not ((flags | self.extra_flags) & F_NOSK):
            ws = self._gemm_workspace(C0.device)
            if ws is not None:
                a.ws, a.ws_bytes = _p(ws), ws.numel() * 4
                a.cu_budget = int(self.sk_cu_budget)
        assert len(groups) <= MAXG
        for i, grp_ in enumerate(groups):
            A, B, Cc = grp_[:3]
            Ng = Cc.shape[-1]
            if mode == GEMM_TN:
                K = A.shape[-2]
                assert A.shape[-1] == M and B.shape[-2] == K and B.shape[-1] == Ng, (A.shape, B.shape, Cc.shape)
            elif mode == GEMM_NN:
                K = A.shape[-1]
                assert A.shape[-2] == M and B.shape[-2] == K and B.shape[-1] == Ng, (A.shape, B.shape, Cc.shape)
            else:
                K = A.shape[-1]
                assert A.shape[-2] == M and B.shape[-1] == K and B.shape[-2] == Ng, (A.shape, B.shape, Cc.shape)
            for t in (A, B, Cc):
                assert t.dtype == torch.float32 and (t.stride(-1) == 1 or t.size(-1) == 1), (t.shape, t.stride())
            assert Cc.shape[-2] == M
            if batched:
                assert (A.stride(0), B.stride(0), Cc.stride(0)) == (a.bsa, a.bsb, a.bsc)
            g = a.g[i]
            g.A, g.B, g.C = _p(A), _p(B), _p(Cc)
            g.lda, g.ldb, g.K, g.N = A.stride(-2), B.stride(-2), K, (Ng if Ng != N else 0)
            g.bias = _p(grp_[3]) if len(grp_) > 3 else None
            g.ldc = Cc.stride(-2) if Cc.stride(-2) != a.ldc else 0
        if plan_only:
            return int(self.lib.dlsg_gemm_variant(C.byref(a)))
        e0 = None
        if self.flop_count is not None:
            self.flop_count += 2.0 * M * nb * sum(grp_[2].shape[-1] * a.g[i].K for i, grp_ in enumerate(groups))
        if self.prof is not None:
            flops = 2.0 * M * nb * sum(grp_[2].shape[-1] * a.g[i].K for i, grp_ in enumerate(groups))     # (groups of different widths)
            if flops >= self.prof_min_flops:     # only the heavy launches are timed, so the events do not perturb the step
                e0 = self._prof_begin()
        self._check(self.lib.dlsg_gemm(C.byref(a), self._stream()), 'dlsg_gemm')
        if e0 is not None:
            x3 = bool(a.flags & F_BF16X3)
            if x3:
                # the split-bf16 dispatch rule (csrc/gemm_bf16x3.hip)
                tiles_l = ((M + 127) // 128) * ((N + 127) // 128) * nb * len(groups)
                if a.flags & (F_FORCE64 | F_FORCE128):
                    variant = '64x64' if a.flags & F_FORCE64 else '128x128'
                elif M <= 64 and mode != GEMM_TN and N >= 64:
                    variant = 'skinny_64x32'
                else:
                    variant = '128x128' if tiles_l >= 1000 else '64x64'
            else:
                v = self.lib.dlsg_gemm_variant(C.byref(a))         # the library's own answer (csrc/gemm.hip, gemm_plan)
                variant = {0: '64x64', 1: '128x64', 2: '128x128', 3: 'skinny_64x32' if M <= 64 else 'skinny_128x32', 4: '256x256',
                           5: '256x128', 6: '256x256+rest', 7: 'streamk_256x256'}[v]
            # one key per kernel symbol (arithmetic, tile, operand layout), as rocprofv3 --stats lists them
            ks = sorted(set(a.g[i].K for i in range(len(groups))))
            ns_ = sorted(set(grp_[2].shape[-1] for grp_ in groups))
            shape = '%s M=%d N=%s K=%s groups=%d batch=%d' % (('NT', 'NN', 'TN')[mode], M, '/'.join(map(str, ns_)), '/'.join(map(str, ks)),
                                                              len(groups), nb)
            # bytes of the operands and results of one launch, every group's counted on its own (groups that share an operand --
            # the two streams' region projections read the same regions -- make the unique bytes smaller than this)
            ob = 0
            for i, grp_ in enumerate(groups):
                Ng = grp_[2].shape[-1]
                Kg = a.g[i].K
                ob += 4 * nb * (M * Kg + Ng * Kg + M * Ng * (2 if (a.flags & F_ACCUM) else 1))
            self._prof_end('gemm_%s_mfma_%s_%s' % ('bf16x3' if x3 else 'f32', variant, ('nt', 'nn', 'tn')[mode]), e0, flops, shape, ob)

    def slab_reduce(self, slabs, out, bias=None, flags=0):
        """slabs (S, rows, n) contiguous per slab; out (rows, n) view."""
        S, rows, n = slabs.shape
        assert slabs.stride(2) == 1 and slabs.stride(1) == n
        self._check(self.lib.dlsg_slab_reduce(_p(slabs), S, i64(slabs.stride(0)), _p(bias), _p(out), i64(rows), n,
                                              int(out.stride(0)), flags | (F_BIAS if bias is not None else 0),
                                              self._stream()), 'dlsg_slab_reduce')

    # ------------------------------------------------------------------ row kernels
    def _rowln_args(self, x, gamma, beta, y, stats, res, pe, pre_tanh, post_tanh, p1, site1, p2, site2, seed, eps):
        a = RowLnArgs()
        _chk2(x)
        a.x, a.ldx = _p(x), x.stride(0)
        a.res, a.ldres = _p(res), (res.stride(0) if res is not None else 0)
        a.gamma, a.beta = _p(gamma), _p(beta)
        a.y, a.ldy = _p(y), (y.stride(0) if y is not None else 0)
        a.stats = _p(stats)
        a.pe, a.pe_rows = _p(pe), (pe.size(0) if pe is not None else 1)
        a.rows, a.n = x.size(0), x.size(1)
        a.pre_tanh, a.post_tanh, a.eps = pre_tanh, post_tanh, eps
        a.p1, a.p2, a.site1, a.site2 = p1, p2, site1, site2
        a.seed, a.seed_ptr = _seed(seed)
        return a

    def rowln_fwd(self, x, gamma, beta, y, stats=None, res=None, pe=None, pre_tanh=0, post_tanh=0, p1=0.0, site1=0,
                  p2=0.0, site2=0, seed=0, eps=1e-5):
        a = self._rowln_args(x, gamma, beta, y, stats, res, pe, pre_tanh, post_tanh, p1, site1, p2, site2, seed, eps)
        self._check(self.lib.dlsg_rowln_fwd(C.byref(a), self._stream()), 'dlsg_rowln_fwd')

    def rowln_fwd_multi(self, items):
        """several norms of the same number of rows in ONE launch; items: dicts of rowln_fwd's arguments (x, gamma, beta, y, ...)"""
        n = len(items)
        if n == 1:
            return self.rowln_fwd(**items[0])
        arr = (RowLnArgs * n)()
        for i, it in enumerate(items):
            d = dict(stats=None, res=None, pe=None, pre_tanh=0, post_tanh=0, p1=0.0, site1=0, p2=0.0, site2=0, seed=0, eps=1e-5)
            d.update(it)
            arr[i] = self._rowln_args(d['x'], d['gamma'], d['beta'], d['y'], d['stats'], d['res'], d['pe'], d['pre_tanh'],
                                      d['post_tanh'], d['p1'], d['site1'], d['p2'], d['site2'], d['seed'], d['eps'])
        self._check(self.lib.dlsg_rowln_fwd_multi(arr, n, self._stream()), 'dlsg_rowln_fwd_multi')

    def rowln_bwd_multi(self, items):
        """several norm backwards of the same number of rows in ONE launch; items: dicts of rowln_bwd's arguments"""
        n = len(items)
        if n == 1:
            return self.rowln_bwd(**items[0])
        arr = (RowLnBwdArgs * n)()
        for i, it in enumerate(items):
            d = dict(stats=None, res=None, pe=None, pre_tanh=0, post_tanh=0, p1=0.0, site1=0, p2=0.0, site2=0, seed=0, eps=1e-5,
                     dgb_part=None, accum_dx=False)
            d.update(it)
            b = arr[i]
            b.f = self._rowln_args(d['x'], d['gamma'], d['beta'], None, d['stats'], d['res'], d['pe'], d['pre_tanh'], d['post_tanh'],
                                   d['p1'], d['site1'], d['p2'], d['site2'], d['seed'], d['eps'])
            b.dy, b.lddy, b.dx, b.lddx = _p(d['dy']), d['dy'].stride(0), _p(d['dx']), d['dx'].stride(0)
            b.accum_dx = int(d['accum_dx'])
            b.dgb_part = _p(d['dgb_part'])
            b.nblk = d['dgb_part'].size(0) if d['dgb_part'] is not None else 0
        self._check(self.lib.dlsg_rowln_bwd_multi(arr, n, self._stream()), 'dlsg_rowln_bwd_multi')

    def rowln_bwd_nblk(self, rows):
        return self.lib.dlsg_rowln_bwd_nblk(int(rows))

    def rowln_bwd(self, dy, x, gamma, beta, dx, stats=None, res=None, pe=None, pre_tanh=0, post_tanh=0, p1=0.0,
                  site1=0, p2=0.0, site2=0, seed=0, eps=1e-5, dgb_part=None, accum_dx=False):
        b = RowLnBwdArgs()
        b.f = self._rowln_args(x, gamma, beta, None, stats, res, pe, pre_tanh, post_tanh, p1, site1, p2, site2, seed,
                               eps)
        b.dy, b.lddy, b.dx, b.lddx = _p(dy), dy.stride(0), _p(dx), dx.stride(0)
        b.accum_dx = int(accum_dx)
        b.dgb_part = _p(dgb_part)
        b.nblk = dgb_part.size(0) if dgb_part is not None else 0
        self._check(self.lib.dlsg_rowln_bwd(C.byref(b), self._stream()), 'dlsg_rowln_bwd')

    def _colsum_ws(self, part):
        """scratch for the fixed-order combine of a tall input's row chunks (None: one chunk, nothing to combine)"""
        k = int(self.lib.dlsg_colsum_ws_floats(part.size(0), part.size(1)))
        return torch.empty(k, dtype=torch.float32, device=part.device) if k else None

    # Short column sums can be collected and launched together (`colsum_defer = []` starts collecting, `colsum_flush()` launches
    # and stops): the backward of a step folds ~25 small partial arrays, each 8-9 us on its own.
    colsum_defer = None

    def _colsum_try_defer(self, part, out_a, out_b, split, dup, accum):
        if self.colsum_defer is None or not self.lib.dlsg_colsum_multi_ok(_p(part), i64(part.stride(0)), part.size(0), part.size(1)):
            return False
        self.colsum_defer.append((part, out_a, out_b, split, dup, accum))
        return True

    def colsum_flush(self, keep_collecting=False):
        """launch the collected column sums: descriptors that write the same destination go into different launches, in order"""
        items, self.colsum_defer = (self.colsum_defer or []), ([] if keep_collecting else None)
        while items:
            batch, rest, seen = [], [], set()
            for it in items:
                dests = {it[1].data_ptr()} | ({it[2].data_ptr()} if it[2] is not None else set())
                if len(batch) < 32 and not (dests & seen) and not rest:
                    batch.append(it); seen |= dests
                else:
                    rest.append(it)
            arr = (ColsumDesc * len(batch))()
            for d, (part, out_a, out_b, split, dup, accum) in zip(arr, batch):
                d.part, d.ld, d.out_a, d.out_b = _p(part), part.stride(0), _p(out_a), _p(out_b)
                d.rows, d.n, d.split, d.dup, d.accum = part.size(0), part.size(1), split, int(dup), int(accum)
            self._check(self.lib.dlsg_colsum_multi(arr, len(batch), self._stream()), 'dlsg_colsum_multi')
            items = rest

    def colsum(self, part, out, accum=False):
        """out[j] (+)= sum over rows of part (rows, n) view."""
        _chk2(part)
        if self._colsum_try_defer(part, out, None, part.size(1), False, accum):
            return
        ws = self._colsum_ws(part)
        self._check(self.lib.dlsg_colsum(_p(part), i64(part.stride(0)), part.size(0), part.size(1), _p(out), int(accum),
                                         _p(ws), self._stream()), 'dlsg_colsum')

    def colsum2(self, part, out_a, out_b, split=None, accum=False):
        """one pass, two destinations: split=k -> columns [0,k) to out_a and [k,n) to out_b; split=None -> every column
        to both out_a and out_b."""
        _chk2(part)
        if self._colsum_try_defer(part, out_a, out_b, 0 if split is None else split, split is None, accum):
            return
        ws = self._colsum_ws(part)
        self._check(self.lib.dlsg_colsum2(_p(part), i64(part.stride(0)), part.size(0), part.size(1), _p(out_a), _p(out_b),
                                          0 if split is None else split, int(split is None), int(accum), _p(ws), self._stream()),
                    'dlsg_colsum2')

    def softmax_fwd(self, x, y, outer, n, inner, mask=None):
        self._check(self.lib.dlsg_softmax_fwd(_p(x), _p(mask), _p(y), i64(outer), n, inner, self._stream()), 'softmax_fwd')

    def softmax_bwd(self, y, dy, dx, outer, n, inner):
        self._check(self.lib.dlsg_softmax_bwd(_p(y), _p(dy), _p(dx), i64(outer), n, inner, self._stream()), 'softmax_bwd')

    # ------------------------------------------------------------------ o2v graph
    def o2v_supported(self, T, H):
        return T <= 32 and H in (64, 512, 1024)

    def o2v_fwd(self, y, v, g_obj, b_obj, z, ml, ostats, S, scale, nsplit, eps=1e-5):
        self.o2v_fwd_multi([dict(y=y, v=v, g_obj=g_obj, b_obj=b_obj, z=z, ml=ml, ostats=ostats, S=S)], scale, nsplit, eps)

    def o2v_fwd_multi(self, items, scale, nsplit, eps=1e-5):
        """Several object->frame graphs of one shape in ONE launch (the object and the motion stream of CapGnnEncoder).
        items: dicts with y (B,NO,H), v (B,T,H), g_obj, b_obj, z, ml, ostats, S."""
        n = len(items)
        B, NO, H = items[0]['y'].shape
        T = items[0]['v'].shape[1]
        arr = (O2VArgs * n)()
        wsb = self.lib.dlsg_o2v_workspace_bytes(B, T, H, nsplit)
        keep = []
        for a, it in zip(arr, items):
            assert it['y'].shape == (B, NO, H) and it['v'].shape[1] == T
            ws = torch.empty(wsb // 4, dtype=torch.float32, device=it['y'].device)
            keep.append(ws)
            a.y, a.v, a.g_obj, a.b_obj, a.z, a.ml, a.ostats, a.S = _p(it['y']), _p(it['v']), _p(it['g_obj']), _p(it['b_obj']), \
                _p(it['z']), _p(it['ml']), _p(it['ostats']), _p(it['S'])
            a.ws, a.ws_bytes = _p(ws), wsb
            a.B, a.T, a.NO, a.H, a.nsplit, a.scale, a.eps = B, T, NO, H, nsplit, scale, eps
        e0 = self._prof_begin()
        self._check(self.lib.dlsg_o2v_fwd_multi(arr, n, self._stream()), 'dlsg_o2v_fwd_multi')
        # algorithmic bytes: read y once + read v + write z, per graph
        self._prof_end('o2v_graph_fwd', e0, 4.0 * n * B * (NO * H + 2 * T * H), 'B=%d T=%d NO=%d H=%d streams=%d nsplit=%d' % (B, T, NO, H, n, nsplit))

    def o2v_bwd(self, y, ostats, g_obj, b_obj, v, z, dz, S, ml, dy, dv, scale, nsplit):
        """backward of o2v_fwd: dz (B,T,H) -> dy (B,NO,H), dv (B,T,H); returns part (B*nsplit,2,H) (obj_norm dgamma | dbeta per
        (clip, object chunk))."""
        return self.o2v_bwd_multi([dict(y=y, ostats=ostats, g_obj=g_obj, b_obj=b_obj, v=v, z=z, dz=dz, S=S, ml=ml, dy=dy, dv=dv)],
                                  scale, nsplit)[0]

    def o2v_bwd_multi(self, items, scale, nsplit):
        """the backward of several object->frame graphs of one shape, one launch per pass (both encoder streams).
        items: dicts with y (B,NO,H), ostats, g_obj, b_obj, v, z, dz (B,T,H), S, ml, dy, dv and optionally dysum (B*nsplit,H): the
        column sums of dy per (clip, object chunk).  Returns the list of `part` arrays."""
        n = len(items)
        B, NO, H = items[0]['y'].shape
        T = items[0]['v'].shape[1]
        dev = items[0]['y'].device
        arr = (O2VBwdArgs * n)()
        wsb = int(self.lib.dlsg_o2v_workspace_bytes(B, T, H, nsplit))
        keep, parts = [], []
        for a, it in zip(arr, items):
            for k in ('y', 'ostats', 'v', 'z', 'dz', 'S', 'ml', 'dy', 'dv'):
                _chkc(it[k])
            assert it['y'].shape == (B, NO, H) and it['v'].shape[1] == T
            pd = torch.empty(B, NO, 64, dtype=torch.float32, device=dev)
            m12 = torch.empty(B, NO, 2, dtype=torch.float32, device=dev)
            ws = torch.empty(wsb // 4, dtype=torch.float32, device=dev)
            part = torch.empty(B * nsplit, 2, H, dtype=torch.float32, device=dev)
            keep += [pd, m12, ws]
            parts.append(part)
            for k in ('y', 'ostats', 'g_obj', 'b_obj', 'v', 'z', 'dz', 'S', 'ml', 'dy', 'dv'):
                setattr(a, k, _p(it[k]))
            a.pd, a.m12, a.part, a.ws, a.ws_bytes = _p(pd), _p(m12), _p(part), _p(ws), wsb
            if it.get('dysum') is not None:
                _chkc(it['dysum'])
                assert it['dysum'].shape == (B * nsplit, H)
                a.dysum = _p(it['dysum'])
            a.B, a.T, a.NO, a.H, a.nsplit, a.scale = B, T, NO, H, nsplit, scale
        e0 = self._prof_begin()
        self._check(self.lib.dlsg_o2v_bwd_multi(arr, n, self._stream()), 'dlsg_o2v_bwd_multi')
        # algorithmic bytes: y read by both passes is counted once (SURVEY.md 8d convention) + dy written + dz, v, dv
        self._prof_end('o2v_graph_bwd', e0, 4.0 * n * B * (2 * NO * H + 3 * T * H), 'B=%d T=%d NO=%d H=%d streams=%d nsplit=%d' % (B, T, NO, H, n, nsplit))
        return parts

    # ------------------------------------------------------------------ beam search
    def beam_select(self, logits, last, last_lp, pred, new_lp, back, rows, k, end, first=False, ended_count=None):
        """one beam-search step for every batch item (see include/dlsg.h): logits (B*k,V) -> pred/new_lp/back/rows (B*k)."""
        a = BeamSelectArgs()
        R, V = logits.shape
        a.logits, a.ld = _p(logits), logits.stride(0)
        a.last, a.last_lp = _p(last), _p(last_lp)
        a.pred, a.new_lp, a.back, a.rows, a.ended_count = _p(pred), _p(new_lp), _p(back), _p(rows), _p(ended_count)
        a.B, a.k, a.V, a.end, a.first = R // k, k, V, end, int(first)
        self._check(self.lib.dlsg_beam_select(C.byref(a), self._stream()), 'dlsg_beam_select')

    def gather_rows_multi(self, srcs, rows, dsts):
        """dsts[i][r] = srcs[i][rows[r]] for up to 4 dense (R, n_i) arrays in one launch."""
        a = GatherMultiArgs()
        for i, (sr, ds) in enumerate(zip(srcs, dsts)):
            _chkc(sr); _chkc(ds)
            a.src[i], a.dst[i], a.n[i] = sr.data_ptr(), ds.data_ptr(), sr.shape[1]
        a.rows, a.nrows, a.count = _p(rows), rows.numel(), len(srcs)
        self._check(self.lib.dlsg_gather_rows_multi(C.byref(a), self._stream()), 'dlsg_gather_rows_multi')

    # ------------------------------------------------------------------ small per-clip graphs of the encoder
    def latent_psl_supported(self, T, P, H):
        return T <= 32 and P <= 32 and H % 4 == 0 and H <= 2048 and T * H * 4 <= 140 * 1024

    def _psl_args(self, a, ov, theta, gamma, beta, adj, u, out, stats, p=0.0, site=0, seed=0, eps=1e-5):
        B, T, H = ov.shape
        P = theta.shape[0]
        for t in (ov, theta, adj, u, out, stats):
            _chkc(t)
        a.ov, a.theta, a.gamma, a.beta, a.adj, a.u, a.out, a.stats = _p(ov), _p(theta), _p(gamma), _p(beta), _p(adj), _p(u), \
            _p(out), _p(stats)
        a.B, a.T, a.P, a.H, a.p, a.site, a.eps = B, T, P, H, p, site, eps
        a.seed, a.seed_ptr = _seed(seed)

    def latent_psl_fwd(self, ov, theta, gamma, beta, adj, u, out, stats, p=0.0, site=0, seed=0, eps=1e-5):
        """ov (B,T,H), theta (P,H) -> adj (B,T,P), u (B*P,H) pre-activation, out (B*P,H), stats (B*P,2); one launch."""
        a = LatentPslArgs()
        self._psl_args(a, ov, theta, gamma, beta, adj, u, out, stats, p, site, seed, eps)
        self._check(self.lib.dlsg_latent_psl_fwd(C.byref(a), self._stream()), 'dlsg_latent_psl_fwd')

    def latent_psl_fwd_multi(self, items):
        """several LatentPSL modules of one shape in ONE launch; items: dicts of latent_psl_fwd's arguments"""
        n = len(items)
        arr = (LatentPslArgs * n)()
        for i, it in enumerate(items):
            self._psl_args(arr[i], **it)
        self._check(self.lib.dlsg_latent_psl_fwd_multi(arr, n, self._stream()), 'dlsg_latent_psl_fwd_multi')

    def latent_psl_bwd_supported(self, T, P, H):
        return T <= 32 and P <= 8 and H % 4 == 0 and H <= 2048 and (T + 8) * H * 4 <= 150 * 1024

    def _psl_bwd_args(self, a, dout, u, stats, gamma, adj, ov, theta, dov, dtheta_part, part, p=0.0, site=0, seed=0):
        B, T, H = ov.shape
        P = theta.shape[0]
        for t in (dout, u, stats, adj, ov, theta, dov, dtheta_part, part):
            _chkc(t)
        a.dout, a.u, a.stats, a.gamma, a.adj, a.ov, a.theta = _p(dout), _p(u), _p(stats), _p(gamma), _p(adj), _p(ov), _p(theta)
        a.dov, a.dtheta_part, a.part = _p(dov), _p(dtheta_part), _p(part)
        a.B, a.T, a.P, a.H, a.p, a.site = B, T, P, H, p, site
        a.seed, a.seed_ptr = _seed(seed)

    def latent_psl_bwd(self, dout, u, stats, gamma, adj, ov, theta, dov, dtheta_part, part, p=0.0, site=0, seed=0):
        """backward of latent_psl_fwd: dout (B*P,H) -> dov (B*T,H), dtheta_part (B,P,H), part (B,2,H); one launch."""
        a = LatentPslBwdArgs()
        self._psl_bwd_args(a, dout, u, stats, gamma, adj, ov, theta, dov, dtheta_part, part, p, site, seed)
        self._check(self.lib.dlsg_latent_psl_bwd(C.byref(a), self._stream()), 'dlsg_latent_psl_bwd')

    def latent_psl_bwd_multi(self, items):
        """several LatentPSL backwards of one shape in ONE launch; items: dicts of latent_psl_bwd's arguments"""
        n = len(items)
        arr = (LatentPslBwdArgs * n)()
        for i, it in enumerate(items):
            self._psl_bwd_args(arr[i], **it)
        self._check(self.lib.dlsg_latent_psl_bwd_multi(arr, n, self._stream()), 'dlsg_latent_psl_bwd_multi')

    def sa_core_supported(self, T, D):
        return T <= 32 and D % 64 == 0

    def sa_core_fwd(self, K, Q, V, w, out, scale, mask=None):
        """K, Q, V (B,T,D) -> w (B,T,T) = softmax_j(K_i.Q_j*scale), out (B,T,D) = w V; one launch."""
        B, T, D = K.shape
        for t in (K, Q, V, w, out):
            _chkc(t)
        a = SaCoreArgs()
        a.K, a.Q, a.V, a.mask, a.w, a.out = _p(K), _p(Q), _p(V), _p(mask), _p(w), _p(out)
        a.B, a.T, a.D, a.scale = B, T, D, scale
        self._check(self.lib.dlsg_sa_core_fwd(C.byref(a), self._stream()), 'dlsg_sa_core_fwd')

    def sa_core_bwd(self, w, K, Q, V, dout, dK, dQ, dV, scale):
        """backward of sa_core_fwd: dout (B,T,D) + saved w (B,T,T) -> dK, dQ, dV (B,T,D); one launch."""
        B, T, D = K.shape
        for t in (w, K, Q, V, dout, dK, dQ, dV):
            _chkc(t)
        a = SaCoreBwdArgs()
        a.w, a.K, a.Q, a.V, a.dout, a.dK, a.dQ, a.dV = _p(w), _p(K), _p(Q), _p(V), _p(dout), _p(dK), _p(dQ), _p(dV)
        a.B, a.T, a.D, a.scale = B, T, D, scale
        self._check(self.lib.dlsg_sa_core_bwd(C.byref(a), self._stream()), 'dlsg_sa_core_bwd')

    # ------------------------------------------------------------------ decoder attention
    def _decatt_args(self, Kp, Vp, q, c, alpha, scale):
        a = DecAttArgs()
        ns = len(Kp)
        B, P, Q = Kp[0].shape
        H = Vp[0].shape[2]
        for s in range(ns):
            a.Kp[s], a.Vp[s] = Kp[s].data_ptr(), Vp[s].data_ptr()
            if c is not None:
                a.c[s] = c[s].data_ptr()
        a.q, a.ldq = _p(q), q.stride(0)
        a.ldc = c[0].stride(0) if c is not None else 0
        a.alpha = _p(alpha)
        a.B, a.P, a.Q, a.H, a.nstream, a.scale = B, P, Q, H, ns, scale
        return a

    def decatt_fwd(self, Kp, Vp, q, c, alpha, scale):
        a = self._decatt_args(Kp, Vp, q, c, alpha, scale)
        self._check(self.lib.dlsg_decatt_fwd(C.byref(a), self._stream()), 'dlsg_decatt_fwd')

    def decatt_bwd(self, Kp, Vp, q, alpha, dc, dKp, dVp, dq, scale, accum_dq=False, dalpha=None):
        b = DecAttBwdArgs()
        b.f = self._decatt_args(Kp, Vp, q, None, alpha, scale)
        for s in range(len(Kp)):
            b.dc[s], b.dKp[s], b.dVp[s] = dc[s].data_ptr(), dKp[s].data_ptr(), dVp[s].data_ptr()
        b.lddc = dc[0].stride(0)
        b.dalpha = _p(dalpha)
        b.dq, b.lddq, b.accum_dq = _p(dq), dq.stride(0), int(accum_dq)
        self._check(self.lib.dlsg_decatt_bwd(C.byref(b), self._stream()), 'dlsg_decatt_bwd')

    # ------------------------------------------------------------------ fused decoder step
    def dec_mid_fwd(self, slabs, addend, b_ih, b_hh, c_prev, c, h, gates, lnq, qcur, st_q, p_q, site_q, Kp, Vp, lnc, cpre,
                    ctx, st_c, alpha, p_att, site_att, scale, seed=0, eps=1e-5, kv_div=1):
        """query cell pointwise -> LN(+dropout) -> attention over Kp/Vp (per stream) -> tanh -> LN(+dropout); one launch.
        lnq = (gamma, beta); lnc = [(gamma, beta)] per stream.  kv_div = k > 1: Kp / Vp hold one block per k consecutive rows
        (beam search: the k beams of a clip share their clip's K', V')."""
        a = DecMidArgs()
        a.kv_div = kv_div
        assert Kp[0].size(0) * max(1, kv_div) == c.size(0), (Kp[0].shape, c.shape, kv_div)
        a.slabs, a.nslab, a.slab_stride = _p(slabs), slabs.size(0), slabs.stride(0)
        a.addend, a.ldadd = _p(addend), (addend.stride(0) if addend is not None else 0)
        a.b_ih, a.b_hh, a.c_prev, a.c, a.h, a.gates = _p(b_ih), _p(b_hh), _p(c_prev), _p(c), _p(h), _p(gates)
        a.lnq_g, a.lnq_b, a.qcur, a.st_q, a.p_q, a.site_q = _p(lnq[0]), _p(lnq[1]), _p(qcur), _p(st_q), p_q, site_q
        ns = len(Kp)
        for i in range(ns):
            _chkc(Kp[i]); _chkc(Vp[i]); _chkc(cpre[i]); _chkc(ctx[i])
            a.Kp[i], a.Vp[i] = Kp[i].data_ptr(), Vp[i].data_ptr()
            a.lnc_g[i], a.lnc_b[i] = lnc[i][0].data_ptr(), lnc[i][1].data_ptr()
            a.cpre[i], a.ctx[i], a.st_c[i] = cpre[i].data_ptr(), ctx[i].data_ptr(), st_c[i].data_ptr()
            a.p_att[i], a.site_att[i] = p_att[i], site_att[i]
        for t in (c, h, gates, qcur, alpha):
            _chkc(t)
        a.alpha = _p(alpha)
        a.B, a.Q, a.H, a.P, a.nstream = c.size(0), c.size(1), Vp[0].size(2), Kp[0].size(1), ns
        a.scale, a.eps = scale, eps
        a.seed, a.seed_ptr = _seed(seed)
        self._check(self.lib.dlsg_dec_mid_fwd(C.byref(a), self._stream()), 'dlsg_dec_mid_fwd')

    # vocabulary x width up to which a workgroup projects its own row (dec_tail_fwd sample=).  The in-launch projection sums a row's
    # dot products in another order than the vocabulary GEMM whose logits the loss sees (and than the unfused path above this
    # size: MSR-VTT's 10 000 x 1024): two logits closer than ~1e-6 relative can order differently, so at an exact near-tie the
    # sampled word may differ from argmax(logits).  The reference itself is not bit-stable there (its GEMM's order is the
    # library's); every fixture with sampled steps -- reference-generated, both sides of this threshold -- gives identical ids
    # (tests/test_gpu_parity.py, test_gpu_bench_parity.py: MSVD below, MSR-VTT above).
    DEC_TAIL_SAMPLE_MAX_WEIGHTS = 1 << 21

    def dec_tail_sample_supported(self, V, D):
        return V * D <= self.DEC_TAIL_SAMPLE_MAX_WEIGHTS

    def dec_tail_fwd(self, slabs, b_ih, b_hh, c_prev, c, hd, gates, ln, dout, st_l, p, site, seed=0, eps=1e-5, sample=None):
        """language cell pointwise (+dropout on h) -> tanh(LN(h)); one launch.
        sample: optional dict(coins (int32 device vector), t, W (V,D), b (V) or None, E (V,Wd), ids_out (B) int64, we_out (B,Wd),
        p, site, row0): the next step's word is sampled inside the launch when coins[t] == 0 (include/dlsg.h)."""
        a = DecTailArgs()
        if sample is not None:
            sm = sample
            for t_ in (sm['W'], sm['E'], sm['ids_out']):
                _chkc(t_)
            assert sm['W'].shape[1] == c.size(1) and sm['we_out'].stride(1) == 1 and sm['ids_out'].dtype == torch.int64
            a.s_coins, a.s_t, a.s_V = _p(sm['coins']), int(sm['t']), sm['W'].shape[0]
            a.s_W, a.s_b, a.s_E, a.s_Wd = _p(sm['W']), _p(sm.get('b')), _p(sm['E']), sm['E'].shape[1]
            a.s_site, a.s_ids, a.s_we, a.s_ldwe = sm.get('site', 0), _p(sm['ids_out']), _p(sm['we_out']), sm['we_out'].stride(0)
            a.s_row0, a.s_p = int(sm.get('row0', 0)), float(sm.get('p', 0.0))
        a.slabs, a.nslab, a.slab_stride = _p(slabs), slabs.size(0), slabs.stride(0)
        a.b_ih, a.b_hh, a.c_prev, a.c, a.hd, a.gates = _p(b_ih), _p(b_hh), _p(c_prev), _p(c), _p(hd), _p(gates)
        for t in (c, hd, gates, dout):
            _chkc(t)
        a.ln_g, a.ln_b, a.dout, a.st_l, a.p, a.site = _p(ln[0]), _p(ln[1]), _p(dout), _p(st_l), p, site
        a.B, a.D, a.eps = c.size(0), c.size(1), eps
        a.seed, a.seed_ptr = _seed(seed)
        self._check(self.lib.dlsg_dec_tail_fwd(C.byref(a), self._stream()), 'dlsg_dec_tail_fwd')

    def dec_mid_bwd(self, slabs, dlh_rec, cpre, st_c, lnc_g, part_c, dcpre, p_att, site_att, Kp, Vp, alpha, dalpha, ds, qh,
                    st_q, lnq_g, part_q, p_q, site_q, rec_slabs, gates, c, c_prev, dc, dgates, scale, seed=0):
        """backward of dec_mid_fwd for one word step (see include/dlsg.h).  slabs (S,B,ns*H+Q+D); dlh_rec (B,D) or None;
        rec_slabs (S',B,Q) view of the query cell's input-gradient slabs of step t+1, or None."""
        a = DecMidBwdArgs()
        ns = len(Kp)
        B, Q = c.shape
        H, P = Vp[0].size(2), Kp[0].size(1)
        D = slabs.size(2) - ns * H - Q
        _chkc(slabs)
        a.slabs, a.nslab, a.slab_stride = _p(slabs), slabs.size(0), slabs.stride(0)
        a.write_rec, a.dlh_rec = int(dlh_rec is not None), _p(dlh_rec)
        for i in range(ns):
            for t in (cpre[i], part_c[i], dcpre[i], Kp[i], Vp[i]):
                _chkc(t)
            a.cpre[i], a.st_c[i], a.lnc_g[i] = cpre[i].data_ptr(), st_c[i].data_ptr(), lnc_g[i].data_ptr()
            a.part_c[i], a.dcpre[i] = part_c[i].data_ptr(), dcpre[i].data_ptr()
            a.p_att[i], a.site_att[i] = p_att[i], site_att[i]
            a.Kp[i], a.Vp[i] = Kp[i].data_ptr(), Vp[i].data_ptr()
        for t in (alpha, ds, qh, part_q, gates, c, dc, dgates) + ((dlh_rec,) if dlh_rec is not None else ()):
            _chkc(t)
        a.alpha, a.dalpha, a.ds = _p(alpha), _p(dalpha), _p(ds)
        a.qh, a.st_q, a.lnq_g, a.part_q, a.p_q, a.site_q = _p(qh), _p(st_q), _p(lnq_g), _p(part_q), p_q, site_q
        if rec_slabs is not None:
            a.rec_slabs, a.rec_nslab = _p(rec_slabs), rec_slabs.size(0)
            a.rec_slab_stride, a.rec_ld = rec_slabs.stride(0), rec_slabs.stride(1)
        a.gates, a.c, a.c_prev, a.dc, a.dgates = _p(gates), _p(c), _p(c_prev), _p(dc), _p(dgates)
        a.B, a.Q, a.H, a.D, a.P, a.nstream, a.scale = B, Q, H, D, P, ns, scale
        a.seed, a.seed_ptr = _seed(seed)
        self._check(self.lib.dlsg_dec_mid_bwd(C.byref(a), self._stream()), 'dlsg_dec_mid_bwd')

    def decatt_cache_grads(self, alpha, ds, qcur, dcpre, dKp, dVp):
        """dK'[s] = sum_t ds_t (x) q_cur_t, dV'[s] = sum_t alpha_t (x) dcpre_t  (time-major (L,B,.) inputs)."""
        a = DecattCacheGradsArgs()
        L, B, Q = qcur.shape
        ns = len(dKp)
        for t in (alpha, ds, qcur):
            _chkc(t)
        a.alpha, a.ds, a.qcur = _p(alpha), _p(ds), _p(qcur)
        for i in range(ns):
            for t in (dcpre[i], dKp[i], dVp[i]):
                _chkc(t)
            a.dcpre[i], a.dKp[i], a.dVp[i] = dcpre[i].data_ptr(), dKp[i].data_ptr(), dVp[i].data_ptr()
        a.L, a.B, a.Q, a.H, a.P, a.nstream = L, B, Q, dVp[0].size(2), dKp[0].size(1), ns
        self._check(self.lib.dlsg_decatt_cache_grads(C.byref(a), self._stream()), 'dlsg_decatt_cache_grads')

    # ------------------------------------------------------------------ persistent kernels: time-out word
    def _persist_word(self, dev):
        """the int32 device word every persistent recurrent launch of this process reports a hand-off time-out into
        (csrc/bilstm.hip, csrc/critic_lstm.hip) and dlsg_adam reads as its guard"""
        w = getattr(self, '_persist_err', None)
        if w is None or w.device != dev:
            w = self._persist_err = torch.zeros(1, dtype=torch.int32, device=dev)
        return w

    guard_word = _persist_word        # (public name: the word a Trainer with several ranks max-reduces in front of Adam)

    def persist_word_or_none(self):
        """the time-out word (int32 device tensor) if a persistent kernel ran in this process: a caller that reads scalars back
        anyway appends it to that read and hands the value to `check_persistent(code=...)` -- one host synchronisation, not two"""
        return getattr(self, '_persist_err', None)

    def check_persistent(self, code=None):
        """Read the persistent kernels' time-out word: a workgroup that waited ~1 s for another one's flag (the launch was not
        fully co-resident: the device is shared with another process, or another stream held the compute units) sets it; results
        since then are invalid and every Adam launch since then was a no-op (dlsg_adam's guard).  Raises, after switching the
        BiLSTM to its step-by-step schedule for later launches.  A host synchronisation: called where a loss or token ids are read
        back anyway (Trainer.step every `check_every` steps, GanTrainer.iteration, beam.beam_finish, scoring.gather_results)."""
        w = getattr(self, '_persist_err', None)
        if w is None:
            return
        if code is None:
            code = int(w.item())
        if code:
            w.zero_()
            self.persistent_bilstm = False
            # graphs captured so far replay the persistent launches: their owners (Trainer, GanTrainer) compare this count and
            # capture again on the step-by-step schedule
            self.persist_timeouts = getattr(self, 'persist_timeouts', 0) + 1
            raise RuntimeError('persistent kernel hand-off timed out (code %d): the launch was not co-resident on this device.  '
                               'No parameter was updated by the steps since; the BiLSTM now runs step by step '
                               '(ops.persistent_bilstm = False)%s.  The critic\'s LSTM has no step-by-step form: GAN training needs '
                               'the device to itself.  Is the GPU shared with another process?' % (code, ''))

    # ------------------------------------------------------------------ persistent BiLSTM recurrence
    persistent_bilstm = True      # False: the per-step schedule (grouped skinny GEMM + pointwise launch per step)
    persistent_bilstm_bwd = True  # False: only the backward through time step by step (Trainer sets it when world_size > 1)

    def bilstm_supported(self, B, T, H):
        return self.persistent_bilstm and bool(self.lib.dlsg_bilstm_supported(B, T, H))

    def bilstm_fwd(self, xg, w_hh, b_ih, b_hh, out, hprev, c, gates):
        """all T steps of both directions in one launch (csrc/bilstm.hip).  xg[d] (B*T, 4H) rows b*T+t; out (B,T,2H);
        hprev[d] (B,T,H) zero-filled by the caller; c[d] (B,T,H); gates[d] (B,T,4H).  Returns the int32 error word (device)."""
        B, T, H2 = out.shape
        H = H2 // 2
        a = BilstmArgs()
        dev = out.device
        hx = torch.empty(int(self.lib.dlsg_bilstm_hx_floats(T, H)), dtype=torch.float32, device=dev)
        flags = torch.empty(int(self.lib.dlsg_bilstm_flag_words(T, H)), dtype=torch.int32, device=dev)
        self._bilstm_err = self._persist_word(dev)
        for d in range(2):
            for t in (xg[d], w_hh[d], hprev[d], c[d], gates[d]):
                _chkc(t) if t is not xg[d] else _chk2(t)
            a.xg[d], a.w_hh[d], a.b_ih[d], a.b_hh[d] = xg[d].data_ptr(), w_hh[d].data_ptr(), b_ih[d].data_ptr(), b_hh[d].data_ptr()
            a.hprev[d], a.c[d], a.gates[d] = hprev[d].data_ptr(), c[d].data_ptr(), gates[d].data_ptr()
        _chkc(out)
        a.ldxg = xg[0].stride(0)
        assert xg[1].stride(0) == a.ldxg
        a.out, a.hx, a.flags, a.err = _p(out), _p(hx), _p(flags), _p(self._bilstm_err)
        a.B, a.T, a.H = B, T, H
        self._check(self.lib.dlsg_bilstm_fwd(C.byref(a), self._stream()), 'dlsg_bilstm_fwd')
        return self._bilstm_err

    def bilstm_bwd(self, gates, c, dout, w_hh, dgates):
        """backward through time of bilstm_fwd in one launch: dout (B,T,2H) -> dgates[d] (B,T,4H).  Returns the error word."""
        B, T, H2 = dout.shape
        H = H2 // 2
        a = BilstmBwdArgs()
        dev = dout.device
        nx = int(self.lib.dlsg_bilstm_bwd_x_floats(T, H))
        gx = torch.empty(nx, dtype=torch.float32, device=dev)
        px = torch.empty(nx, dtype=torch.float32, device=dev)
        flags = torch.empty(2 * int(self.lib.dlsg_bilstm_flag_words(T, H)), dtype=torch.int32, device=dev)
        self._bilstm_err = self._persist_word(dev)
        _chkc(dout)
        for d in range(2):
            for t in (gates[d], c[d], w_hh[d], dgates[d]):
                _chkc(t)
            a.gates[d], a.c[d], a.w_hh[d], a.dgates[d] = gates[d].data_ptr(), c[d].data_ptr(), w_hh[d].data_ptr(), dgates[d].data_ptr()
        a.dout, a.gx, a.px, a.flags, a.err = _p(dout), _p(gx), _p(px), _p(flags), _p(self._bilstm_err)
        a.B, a.T, a.H = B, T, H
        self._check(self.lib.dlsg_bilstm_bwd(C.byref(a), self._stream()), 'dlsg_bilstm_bwd')
        return self._bilstm_err

    # ------------------------------------------------------------------ critic LSTM, a whole sequence per launch
    def lstm_seq_supported(self, L, n, H):
        """the persistent launches take up to 256 sequences; more run as consecutive launches over caption chunks"""
        return bool(self.lib.dlsg_lstm_seq_supported(L, min(n, 256), H))

    def _lstm_seq(self, level, W, b_ih=None, b_hh=None, **t):
        """batch-major arrays (n, L, .): chunks of <= 256 sequences are independent recurrences, one persistent launch each"""
        ref = next(v for v in t.values() if v is not None)
        n, L = ref.shape[0], ref.shape[1]
        H = W.shape[1]
        dev = W.device
        # sequences per launch: as many row groups of 64 as the device can keep co-resident (256 rows need 4 x H / 8 workgroups
        # = 256 compute units at H = 512; a smaller part takes 192, 128 or 64 rows per launch)
        chunk = next((c for c in (256, 192, 128, 64) if self.lib.dlsg_lstm_seq_supported(L, min(n, c), H)), 0)
        if not chunk:
            raise RuntimeError('dlsg_lstm_seq does not take L = %d, H = %d on this device (H in {64, 512}, >= H / 8 compute units '
                               'per 64 sequences)' % (L, H))
        self._lstm_seq_err = self._persist_word(dev)
        _chkc(W)
        for name, v in t.items():
            if v is not None:
                _chkc(v)
                assert v.dtype == torch.float32 and v.shape[:2] == (n, L) and v.shape[2] in (H, 4 * H), (name, v.shape)
        for lo in range(0, n, chunk):
            hi = min(n, lo + chunk)
            a = LstmSeqArgs()
            nx = int(self.lib.dlsg_lstm_seq_x_floats(L, hi - lo, H))
            xbuf = torch.empty(nx, dtype=torch.float32, device=dev)
            xbuf2 = torch.empty(nx, dtype=torch.float32, device=dev) if level == 1 else None
            flags = torch.empty(int(self.lib.dlsg_lstm_seq_flag_words(L, hi - lo, H)), dtype=torch.int32, device=dev)
            for name, v in t.items():
                setattr(a, name, _p(None if v is None else v[lo:hi]))
            a.W, a.xbuf, a.xbuf2, a.flags, a.err = _p(W), _p(xbuf), _p(xbuf2), _p(flags), _p(self._lstm_seq_err)
            a.b_ih, a.b_hh = _p(b_ih), _p(b_hh)
            a.L, a.n, a.H, a.batch_major = L, hi - lo, H, 1
            self._check(self.lib.dlsg_lstm_seq(C.byref(a), level, self._stream()), 'dlsg_lstm_seq(level %d)' % level)
        return self._lstm_seq_err

    def lstm_seq_fwd(self, xin, W, b_ih, b_hh, As, Hs, Cs, Hprev):
        """h_t, c_t for all L steps in one launch: xin (n, L, 4H) = x W_ih^T; W = weight_hh (4H, H); Hprev[:, t] = h_{t-1}."""
        return self._lstm_seq(0, W, b_ih, b_hh, addend=xin, As=As, Hs=Hs, Cs=Cs, Hprev=Hprev)

    def lstm_seq_bwd(self, As, Cs, W, dHs, dAs, dCs, DA, DH, DC):
        """backward through time with the injected gradients dAs / dCs (None = none) in one launch."""
        return self._lstm_seq(1, W, As=As, Cs=Cs, dHs=dHs, dAs=dAs, dCs=dCs, DA=DA, DH=DH, DC=DC)

    def lstm_seq_bwd2(self, As, Cs, W, DH, DC, ubar, gA, gC, gDH, gDHprev, gDC):
        """derivative of lstm_seq_bwd along ubar (n, L, 4H), the tangent of the input gates, in one launch"""
        return self._lstm_seq(2, W, As=As, Cs=Cs, DH=DH, DC=DC, addend=ubar, gA=gA, gC=gC, gDH=gDH, gDC=gDC, gDHprev=gDHprev)

    # ------------------------------------------------------------------ LSTM pointwise
    @staticmethod
    def _pw_fwd_args(a, slabs, c, B, H, addend=None, b_ih=None, b_hh=None, c_prev=None, h=None, h2=None, gates=None,
                     p=0.0, site=0, seed=0):
        if slabs is not None:
            a.slabs, a.nslab, a.slab_stride = _p(slabs), slabs.size(0), slabs.stride(0)
        else:
            a.slabs, a.nslab, a.slab_stride = None, 0, 0
        a.addend, a.ldadd = _p(addend), (addend.stride(0) if addend is not None else 0)
        a.b_ih, a.b_hh = _p(b_ih), _p(b_hh)
        a.c_prev, a.ldcp = _p(c_prev), (c_prev.stride(0) if c_prev is not None else 0)
        a.c, a.ldc_ = _p(c), c.stride(0)
        a.h, a.ldh = _p(h), (h.stride(0) if h is not None else 0)
        a.h2, a.ldh2 = _p(h2), (h2.stride(0) if h2 is not None else 0)
        a.gates, a.ldg = _p(gates), (gates.stride(0) if gates is not None else 0)
        a.B, a.H, a.p, a.site = B, H, p, site
        a.seed, a.seed_ptr = _seed(seed)

    def lstm_pw_fwd(self, slabs, c, B, H, **kw):
        a = LstmPwArgs()
        self._pw_fwd_args(a, slabs, c, B, H, **kw)
        self._check(self.lib.dlsg_lstm_pw_fwd(C.byref(a), self._stream()), 'dlsg_lstm_pw_fwd')

    def lstm_pw_fwd_multi(self, calls):
        """calls: 1 or 2 dicts of lstm_pw_fwd arguments (same B, H) -> one launch (both directions of a BiLSTM step)."""
        arr = (LstmPwArgs * len(calls))()
        for a, kw in zip(arr, calls):
            self._pw_fwd_args(a, **kw)
        self._check(self.lib.dlsg_lstm_pw_fwd_n(arr, len(calls), self._stream()), 'dlsg_lstm_pw_fwd_n')

    @staticmethod
    def _pw_bwd_args(a, gates, c, dgates, B, H, c_prev=None, dh=None, dh2=None, dc_next=None, dc_prev=None, p=0.0,
                     site=0, seed=0, dh3=None, dh4=None):
        a.gates, a.ldg, a.c, a.ldc_ = _p(gates), gates.stride(0), _p(c), c.stride(0)
        a.c_prev, a.ldcp = _p(c_prev), (c_prev.stride(0) if c_prev is not None else 0)
        a.dh, a.lddh = _p(dh), (dh.stride(0) if dh is not None else 0)
        a.dh2, a.lddh2 = _p(dh2), (dh2.stride(0) if dh2 is not None else 0)
        a.dh3, a.lddh3 = _p(dh3), (dh3.stride(0) if dh3 is not None else 0)
        if dh4 is not None and dh4.dim() == 3:      # slab stack (S,B,H): summed inside the kernel
            a.dh4, a.lddh4, a.dh4_nslab, a.dh4_slab_stride = _p(dh4), dh4.stride(1), dh4.size(0), dh4.stride(0)
        else:
            a.dh4, a.lddh4 = _p(dh4), (dh4.stride(0) if dh4 is not None else 0)
        a.dc_next, a.lddcn = _p(dc_next), (dc_next.stride(0) if dc_next is not None else 0)
        a.dgates, a.lddg = _p(dgates), dgates.stride(0)
        a.dc_prev, a.lddcp = _p(dc_prev), (dc_prev.stride(0) if dc_prev is not None else 0)
        a.B, a.H, a.p, a.site = B, H, p, site
        a.seed, a.seed_ptr = _seed(seed)

    def lstm_pw_bwd(self, gates, c, dgates, B, H, **kw):
        a = LstmPwBwdArgs()
        self._pw_bwd_args(a, gates, c, dgates, B, H, **kw)
        self._check(self.lib.dlsg_lstm_pw_bwd(C.byref(a), self._stream()), 'dlsg_lstm_pw_bwd')

    def lstm_pw_bwd_multi(self, calls):
        arr = (LstmPwBwdArgs * len(calls))()
        for a, kw in zip(arr, calls):
            self._pw_bwd_args(a, **kw)
        self._check(self.lib.dlsg_lstm_pw_bwd_n(arr, len(calls), self._stream()), 'dlsg_lstm_pw_bwd_n')

    # ------------------------------------------------------------------ movers
    def mean_rows_fwd(self, x, out):
        B, P, H = x.shape
        self._check(self.lib.dlsg_mean_rows_fwd(_p(x), _p(out), i64(out.stride(0)), B, P, H, self._stream()), 'mean_rows_fwd')

    def mean_rows_bwd(self, dout, dx, accum=False):
        B, P, H = dx.shape
        self._check(self.lib.dlsg_mean_rows_bwd(_p(dout), i64(dout.stride(0)), _p(dx), B, P, H, int(accum), self._stream()),
                    'mean_rows_bwd')

    def embed_fwd(self, E, ids, out, p=0.0, seed=0, site=0, row0=0):
        rows, W = out.shape
        sd, sp = _seed(seed)
        self._check(self.lib.dlsg_embed_fwd(_p(E), _p(ids), _p(out), i64(out.stride(0)), rows, W, f32(p), u64(sd), u32(site),
                                            i64(row0), sp, self._stream()), 'embed_fwd')

    def select_embed(self, logits, captions, t, coins, E, ids_out, out, p=0.0, seed=0, site=0, row0=0, prefilled=False):
        """ids_out[b] = coins[t] ? captions[b, t] : argmax(logits[b]); out[b] = drop(E[id]).  prefilled: ids_out / out already
        hold the teacher-forced choice (the caller embedded every caption word in one launch): a no-op when coins[t] is set."""
        rows, V = logits.shape
        sd, sp = _seed(seed)
        self._check(self.lib.dlsg_select_embed(_p(logits), i64(logits.stride(0)), V, _p(captions), captions.shape[1], int(t),
                                               _p(coins), _p(E), _p(ids_out), _p(out), i64(out.stride(0)), rows, out.shape[1],
                                               f32(p), u64(sd), u32(site), i64(row0), sp, int(bool(prefilled)), self._stream()), 'select_embed')

    def embed_bwd(self, dout, ids, dE, p=0.0, seed=0, site=0, row0=0):
        rows, W = dout.shape
        sd, sp = _seed(seed)
        self._check(self.lib.dlsg_embed_bwd(_p(dout), i64(dout.stride(0)), _p(ids), _p(dE), rows, W, f32(p), u64(sd),
                                            u32(site), i64(row0), sp, self._stream()), 'embed_bwd')

    def argmax(self, logits, ids):
        rows, V = logits.shape
        self._check(self.lib.dlsg_argmax(_p(logits), i64(logits.stride(0)), _p(ids), rows, V, self._stream()), 'argmax')

    def copy2d(self, src, dst, accum=False):
        rows, n = src.shape
        self._check(self.lib.dlsg_copy2d(_p(src), i64(src.stride(0)), _p(dst), i64(dst.stride(0)), rows, n, int(accum),
                                         self._stream()), 'copy2d')

    def dropout(self, x, y, p, seed, site):
        rows, n = x.shape
        sd, sp = _seed(seed)
        self._check(self.lib.dlsg_dropout(_p(x), i64(x.stride(0)), _p(y), i64(y.stride(0)), rows, n, f32(p), u64(sd),
                                          u32(site), sp, self._stream()), 'dropout')

    def fill(self, t, value):
        assert t.is_contiguous()
        self._check(self.lib.dlsg_fill(_p(t), i64(t.numel()), f32(value), self._stream()), 'fill')

    # ------------------------------------------------------------------ critic schedule blocks (csrc/critic_sched.hip)
    def crit_embed_mix(self, proj_tm, ids, W, bias, eps, h):
        ng, B, L, _ = h.shape
        self._check(self.lib.dlsg_crit_embed_mix(_p(proj_tm), _p(ids), _p(W), _p(bias), _p(eps), _p(h), ng, B, L, W.shape[1],
                                                 self._stream()), 'crit_embed_mix')

    def crit_embed_mix_bwd(self, ch, eps, dhr, dhf_tm):
        ng, B, L, _ = ch.shape
        self._check(self.lib.dlsg_crit_embed_mix_bwd(_p(ch), _p(eps), _p(dhr), _p(dhf_tm), ng, B, L, self._stream()), 'crit_embed_mix_bwd')

    def crit_vocab_scatter(self, dhr, ids, dW):
        _chkc(dhr); _chkc(ids); _chkc(dW)
        self._check(self.lib.dlsg_crit_vocab_scatter(_p(dhr), _p(ids), _p(dW), ids.numel(), dW.shape[1], self._stream()), 'crit_vocab_scatter')

    def crit_relu_taps(self, x, ref, bias, bias_scale, y, taps):
        n, L, _ = x.shape
        for t in (x, ref, y, taps):
            _chkc(t)
        self._check(self.lib.dlsg_crit_relu_taps(_p(x), _p(ref), _p(bias), f32(bias_scale), _p(y), _p(taps), n, L, self._stream()),
                    'crit_relu_taps')

    def crit_relu_taps_bwd(self, dy, dtaps, ref, dx):
        n, L, _ = dy.shape
        for t in (dy, dtaps, ref, dx):
            _chkc(t)
        self._check(self.lib.dlsg_crit_relu_taps_bwd(_p(dy), _p(dtaps), _p(ref), _p(dx), n, L, self._stream()), 'crit_relu_taps_bwd')

    def _cln_args(self, x, gamma, pre_tanh, eps, p_pre, site_pre, p_post, site_post, seed, row0):
        a = ClnArgs()
        G = len(x)
        rows, N = x[0].shape
        for g in range(G):
            _chkc(x[g]); _chkc(gamma[g])
            assert x[g].shape == (rows, N) and x[g].dtype == torch.float32
            a.x[g], a.gamma[g] = _p(x[g]), _p(gamma[g])
        a.rows, a.N, a.groups, a.pre_tanh, a.eps = rows, N, G, int(pre_tanh), eps
        a.p_pre, a.site_pre, a.p_post, a.site_post, a.row0 = p_pre, site_pre, p_post, site_post, row0
        a.seed, a.seed_ptr = _seed(seed)
        a.acc_lo = a.acc_hi = 0
        return a

    def _cln_ws(self, a, dev):
        return torch.empty(a.groups * int(self.lib.dlsg_cln_ws_floats(a.rows, a.N)), dtype=torch.float32, device=dev)

    @staticmethod
    def _cln_dys(a, dys):
        a.ndy = len(dys)
        for k, lst in enumerate(dys):
            for g, t in enumerate(lst):
                _chkc(t)
                a.dy[k][g] = _p(t)

    def cln_fwd(self, x, gamma, beta, y, pre_tanh, eps=1e-5, p_pre=0.0, site_pre=0, p_post=0.0, site_post=0, seed=0, row0=0):
        a = self._cln_args(x, gamma, pre_tanh, eps, p_pre, site_pre, p_post, site_post, seed, row0)
        for g in range(len(x)):
            _chkc(y[g])
            a.beta[g], a.y[g] = _p(beta[g]), _p(y[g])
        self._check(self.lib.dlsg_cln_fwd(C.byref(a), self._stream()), 'cln_fwd')

    def cln_ws_rows(self, rows, N):
        """rows of the per-workgroup partial arrays cln_bwd / cln_bwd2 leave in a deferred workspace (G, 2, this, N)"""
        return int(self.lib.dlsg_cln_ws_floats(rows, N)) // (2 * N)

    def cln_bwd(self, x, gamma, dys, dx, dgamma, dbeta, pre_tanh, eps=1e-5, p_pre=0.0, site_pre=0, p_post=0.0, site_post=0, seed=0,
                row0=0, acc=None, extra=None, defer_ws=None):
        """defer_ws (G, 2, cln_ws_rows, N): the per-workgroup partials of (dgamma, dbeta) are left there, unfolded (the caller sums the
        rows); dgamma / dbeta / extra are then ignored"""
        a = self._cln_args(x, gamma, pre_tanh, eps, p_pre, site_pre, p_post, site_post, seed, row0)
        self._cln_dys(a, dys)
        for g in range(len(x)):
            _chkc(dx[g])
            a.dx[g] = _p(dx[g])
            if dgamma and defer_ws is None:
                a.dgamma[g], a.dbeta[g] = _p(dgamma[g]), _p(dbeta[g])
                if extra is not None:
                    _chkc(extra[g])
                    a.extra[g] = _p(extra[g])
        if acc is not None:
            a.acc_lo, a.acc_hi = acc
        if defer_ws is not None:
            _chkc(defer_ws)
            assert tuple(defer_ws.shape) == (len(x), 2, self.cln_ws_rows(a.rows, a.N), a.N), defer_ws.shape
            ws, a.defer = defer_ws, 1
        else:
            ws = self._cln_ws(a, x[0].device) if dgamma else None
        a.ws = _p(ws)
        self._check(self.lib.dlsg_cln_bwd(C.byref(a), self._stream()), 'cln_bwd')

    def cln_bwd2(self, x, gamma, dys, U, gx, gdy, gpart, pre_tanh, eps=1e-5, p_pre=0.0, site_pre=0, p_post=0.0, site_post=0, seed=0,
                 row0=0, defer_ws=None):
        a = self._cln_args(x, gamma, pre_tanh, eps, p_pre, site_pre, p_post, site_post, seed, row0)
        self._cln_dys(a, dys)
        for g in range(len(x)):
            for t in (U[g], gx[g], gdy[g]):
                _chkc(t)
            a.U[g], a.gx[g], a.gdy[g] = _p(U[g]), _p(gx[g]), _p(gdy[g])
            if defer_ws is None:
                _chkc(gpart[g])
                a.gpart[g] = _p(gpart[g])
        if defer_ws is not None:
            _chkc(defer_ws)
            assert tuple(defer_ws.shape) == (len(x), 2, self.cln_ws_rows(a.rows, a.N), a.N), defer_ws.shape
            ws, a.defer = defer_ws, 1
        else:
            ws = self._cln_ws(a, x[0].device)
        a.ws = _p(ws)
        self._check(self.lib.dlsg_cln_bwd2(C.byref(a), self._stream()), 'cln_bwd2')

    def _sa_args(self, KQV, smask, scale, acc=None):
        a = CritSaArgs()
        _chkc(KQV); _chkc(smask)
        a.KQV, a.smask = _p(KQV), _p(smask)
        a.n, a.L, a.B, a.scale = KQV.shape[0], KQV.shape[1], smask.shape[0], scale
        a.acc_lo, a.acc_hi = acc if acc is not None else (0, 0)
        return a

    def crit_sa_fwd(self, KQV, smask, w, ctx, scale):
        a = self._sa_args(KQV, smask, scale)
        _chkc(w); _chkc(ctx)
        a.w, a.ctx = _p(w), _p(ctx)
        self._check(self.lib.dlsg_crit_sa_fwd(C.byref(a), self._stream()), 'crit_sa_fwd')

    def crit_sa_bwd(self, KQV, smask, w, dctx, dKQV, scale, acc=None):
        a = self._sa_args(KQV, smask, scale, acc)
        for t in (w, dctx, dKQV):
            _chkc(t)
        a.w, a.dctx, a.dKQV = _p(w), _p(dctx), _p(dKQV)
        self._check(self.lib.dlsg_crit_sa_bwd(C.byref(a), self._stream()), 'crit_sa_bwd')

    def crit_sa_bwd2(self, KQV, smask, w, dctx, U, Uctx, gKQV, scale):
        a = self._sa_args(KQV, smask, scale)
        for t in (w, dctx, U, Uctx, gKQV):
            _chkc(t)
        a.w, a.dctx, a.U, a.Uctx, a.gKQV = _p(w), _p(dctx), _p(U), _p(Uctx), _p(gKQV)
        self._check(self.lib.dlsg_crit_sa_bwd2(C.byref(a), self._stream()), 'crit_sa_bwd2')

    @staticmethod
    def _set2(st_, **kw):
        for name, pair in kw.items():
            if pair is None:
                continue
            arr = getattr(st_, name)
            for h in range(2):
                _chkc(pair[h])
                arr[h] = _p(pair[h])

    def _pattn_args(self, a_, e, smask, scale, acc=None):
        a = CritPattnArgs()
        self._set2(a, a=a_, e=e)
        _chkc(smask)
        a.smask = _p(smask)
        a.n, a.L, a.B, a.T, a.scale = a_[0].shape[0], a_[0].shape[1], smask.shape[0], e[0].shape[1], scale
        a.acc_lo, a.acc_hi = acc if acc is not None else (0, 0)
        return a

    def crit_pattn_fwd(self, a_, e, smask, P, wgt, aggpre, scale):
        a = self._pattn_args(a_, e, smask, scale)
        self._set2(a, P=P, wgt=wgt, aggpre=aggpre)
        self._check(self.lib.dlsg_crit_pattn_fwd(C.byref(a), self._stream()), 'crit_pattn_fwd')

    def crit_pattn_bwd(self, a_, e, smask, P, d_agg, d_wgt, da, de, scale, acc=None):
        a = self._pattn_args(a_, e, smask, scale, acc)
        self._set2(a, P=P, d_agg=d_agg, d_wgt=d_wgt, da=da, de=de)
        self._check(self.lib.dlsg_crit_pattn_bwd(C.byref(a), self._stream()), 'crit_pattn_bwd')

    def crit_pattn_bwd2(self, a_, e, smask, P, d_agg, d_wgt, Ua, Uagg, Uwgt, ga, ge, scale):
        a = self._pattn_args(a_, e, smask, scale)
        self._set2(a, P=P, d_agg=d_agg, d_wgt=d_wgt, Ua=Ua, Uagg=Uagg, Uwgt=Uwgt, ga=ga, ge=ge)
        self._check(self.lib.dlsg_crit_pattn_bwd2(C.byref(a), self._stream()), 'crit_pattn_bwd2')

    def _tsum_args(self, words, theta, gamma, beta, fusion, eps, p, site, seed, row0, acc=None):
        a = CritTsumArgs()
        for t in (words, theta, gamma, fusion):
            _chkc(t)
        a.words, a.theta, a.gamma, a.beta, a.fusion = _p(words), _p(theta), _p(gamma), _p(beta), _p(fusion)
        a.n, a.L, a.eps, a.p, a.site, a.row0 = words.shape[0], words.shape[1], eps, p, site, row0
        a.seed, a.seed_ptr = _seed(seed)
        a.acc_lo, a.acc_hi = acc if acc is not None else (0, 0)
        return a

    def crit_tsum_fwd(self, words, theta, gamma, beta, fusion, adj, u, sent, fus, eps=1e-5, p=0.0, site=0, seed=0, row0=0):
        a = self._tsum_args(words, theta, gamma, beta, fusion, eps, p, site, seed, row0)
        for t in (adj, u, sent, fus):
            _chkc(t)
        a.adj, a.u, a.sent, a.fus = _p(adj), _p(u), _p(sent), _p(fus)
        self._check(self.lib.dlsg_crit_tsum_fwd(C.byref(a), self._stream()), 'crit_tsum_fwd')

    def crit_tsum_bwd(self, words, theta, gamma, beta, fusion, adj, u, sent, fus, d_fus, dwords, part, eps=1e-5, p=0.0, site=0, seed=0,
                      row0=0, acc=None):
        """(adj, u, sent, fus of the forward are recomputed by the kernel: accepted for interface symmetry, not read)"""
        a = self._tsum_args(words, theta, gamma, beta, fusion, eps, p, site, seed, row0, acc)
        _chkc(d_fus); _chkc(dwords)
        a.d_fus, a.dwords, a.part = _p(d_fus), _p(dwords), _p(part)
        self._check(self.lib.dlsg_crit_tsum_bwd(C.byref(a), self._stream()), 'crit_tsum_bwd')

    def crit_tsum_bwd2(self, words, theta, gamma, beta, fusion, d_fus, U, Ufus, gwords, gpart, eps=1e-5, p=0.0, site=0, seed=0, row0=0):
        a = self._tsum_args(words, theta, gamma, beta, fusion, eps, p, site, seed, row0)
        for t in (d_fus, U, Ufus, gwords, gpart):
            _chkc(t)
        a.d_fus, a.U, a.Ufus, a.gwords, a.gpart = _p(d_fus), _p(U), _p(Ufus), _p(gwords), _p(gpart)
        self._check(self.lib.dlsg_crit_tsum_bwd2(C.byref(a), self._stream()), 'crit_tsum_bwd2')

    def _score_args(self, v, s, wc, wgt, fus, pair, score, ng, acc=None):
        a = CritScoreArgs()
        self._set2(a, v=v, s=s, wc=wc, wgt=wgt, pair=pair, score=score)
        _chkc(fus)
        a.fus = _p(fus)
        a.n, a.B, a.T, a.ng = s[0].shape[0], v[0].shape[0], v[0].shape[1], ng
        a.acc_lo, a.acc_hi = acc if acc is not None else (0, 0)
        return a

    def crit_score_fwd(self, v, s, wc, bc, wgt, fus, pair, score, both, out, ng):
        a = self._score_args(v, s, wc, wgt, fus, pair, score, ng)
        self._set2(a, bc=bc)
        _chkc(both); _chkc(out)
        a.both, a.out = _p(both), _p(out)
        self._check(self.lib.dlsg_crit_score_fwd(C.byref(a), self._stream()), 'crit_score_fwd')

    def crit_score_bwd(self, v, s, wc, wgt, fus, pair, score, both, d_out, d_fus, c_spre, c_vpre, d_wgt, part_wc, dbc, ng, acc=None):
        a = self._score_args(v, s, wc, wgt, fus, pair, score, ng, acc)
        self._set2(a, c_spre=c_spre, c_vpre=c_vpre, d_wgt=d_wgt, part_wc=part_wc)
        for t in (both, d_out, d_fus):
            _chkc(t)
        a.both, a.d_out, a.d_fus, a.dbc = _p(both), _p(d_out), _p(d_fus), _p(dbc)
        scratch = torch.empty(4 * ng + 16, dtype=torch.float32, device=fus.device)
        a.scratch = _p(scratch)
        self._check(self.lib.dlsg_crit_score_bwd(C.byref(a), self._stream()), 'crit_score_bwd')

    def crit_score_bwd2(self, v, s, wc, bc, wgt, fus, pair, score, d_out, Uspre, Uwgt, Ufus, g_fus, g_spre, g_vpre, g_wgt, gpart_wc, g_dbc):
        a = self._score_args(v, s, wc, wgt, fus, pair, score, 1)
        self._set2(a, Uspre=Uspre, Uwgt=Uwgt, c_spre=g_spre, c_vpre=g_vpre, d_wgt=g_wgt, part_wc=gpart_wc)
        for t in (d_out, Ufus, g_fus, g_dbc):
            _chkc(t)
        a.d_out, a.Ufus, a.d_fus, a.dbc = _p(d_out), _p(Ufus), _p(g_fus), _p(g_dbc)
        n = s[0].shape[0]
        scratch = torch.empty(16 + 2 * n + 2 * n * 8, dtype=torch.float32, device=fus.device)
        a.scratch = _p(scratch)
        self._check(self.lib.dlsg_crit_score_bwd2(C.byref(a), self._stream()), 'crit_score_bwd2')

    def crit_gp(self, g, gG, out, stats, vseed, gsc):
        B, L, _ = g.shape
        for t in (g, gG, out, stats, vseed, gsc):
            _chkc(t)
        q = torch.empty(B, dtype=torch.float32, device=g.device)
        self._check(self.lib.dlsg_crit_gp(_p(g), _p(gG), _p(out), _p(stats), _p(vseed), _p(gsc), _p(q), B, L, self._stream()), 'crit_gp')

    def crit_topk(self, alpha, smask, P, T, idx):
        B, L, na = alpha.shape
        assert alpha.stride(2) == 1 and alpha.dtype == torch.float32
        _chkc(smask); _chkc(idx)
        self._check(self.lib.dlsg_crit_topk(_p(alpha), i64(alpha.stride(0)), i64(alpha.stride(1)), na, _p(smask), _p(idx), B, L, P, T,
                                            self._stream()), 'crit_topk')

    def crit_unselect(self, src, idx, dst, per):
        """dst (R, n): row idx[r] = src[r], every other row zero; dst is groups of `per` rows, idx holds the selected rows of each
        group (the same number per group), group by group"""
        _chkc(src); _chkc(idx); _chkc(dst)
        self._check(self.lib.dlsg_crit_unselect(_p(src), _p(idx), _p(dst), src.shape[0], dst.shape[0], per, dst.shape[1], self._stream()),
                    'crit_unselect')

    def crit_reduce(self, descs):
        """descs: list of (slabs (S, ...) contiguous, out): out = sum over the S slabs, in slab order; one launch per 16"""
        for lo in range(0, len(descs), 16):
            chunk = descs[lo:lo + 16]
            arr = (CritReduceDesc * len(chunk))()
            for d, (slabs, out) in zip(arr, chunk):
                _chkc(slabs); _chkc(out)
                assert out.numel() * slabs.shape[0] == slabs.numel() and out.numel() % 4 == 0
                d.src, d.stride, d.n, d.nslab, d.scale, d.out = _p(slabs), out.numel(), out.numel(), slabs.shape[0], 1.0, _p(out)
            self._check(self.lib.dlsg_crit_reduce(arr, len(chunk), self._stream()), 'crit_reduce')

    def crit_colsum(self, descs):
        """descs: list of (sources, out, out_b, scale): out = scale * sum over the rows of the one or two 2-d sources (unit column
        stride); out_b (or None) receives a copy.  One launch per 48 descriptors, fixed order of additions."""
        for lo in range(0, len(descs), 48):
            chunk = descs[lo:lo + 48]
            arr = (CritColsumDesc * len(chunk))()
            for d, (srcs, out, out_b, scale) in zip(arr, chunk):
                a = srcs[0]
                assert a.dim() == 2 and (a.stride(1) == 1 or a.shape[1] == 1) and out.numel() == a.shape[1] and out.is_contiguous()
                d.part, d.ld, d.rows, d.n = _p(a), a.stride(0), a.shape[0], a.shape[1]
                if len(srcs) > 1:
                    b = srcs[1]
                    assert b.dim() == 2 and (b.stride(1) == 1 or b.shape[1] == 1) and b.shape[1] == a.shape[1]
                    d.part_b, d.ld_b, d.rows_b = _p(b), b.stride(0), b.shape[0]
                d.out, d.out_b, d.scale = _p(out), _p(out_b), scale
            self._check(self.lib.dlsg_crit_colsum(arr, len(chunk), self._stream()), 'crit_colsum')

    def gather_rows(self, src, idx, dst):
        """dst[r] = src[idx[r]] (2-d views; dst must not alias src)."""
        rows, n = dst.shape
        self._check(self.lib.dlsg_gather_rows(_p(src), i64(src.stride(0)), _p(idx), _p(dst), i64(dst.stride(0)), rows, n,
                                              self._stream()), 'gather_rows')

    def permute_tb(self, src, dst):
        """dst[b, t, :] = src[t, b, :]  (both contiguous 3-d)."""
        T, B, n = src.shape
        self._check(self.lib.dlsg_permute_tb(_p(src), _p(dst), T, B, n, self._stream()), 'permute_tb')

    # ------------------------------------------------------------------ loss / optimizer
    def ce_ragged(self, logits, targets, lens, dlogits, row_loss, loss, time_major):
        """logits (B,L,V) or, time_major, (L,B,V); targets (B,L) int64; lens (B,) int64."""
        if time_major:
            L, B, V = logits.shape
        else:
            B, L, V = logits.shape
        self._check(self.lib.dlsg_ce_ragged(_p(logits), _p(targets), _p(lens), _p(dlogits), _p(row_loss), _p(loss), B, L, V,
                                            int(time_major), self._stream()), 'ce_ragged')

    def log_softmax(self, logits, out):
        rows, V = logits.shape
        self._check(self.lib.dlsg_log_softmax(_p(logits), _p(out), rows, V, self._stream()), 'log_softmax')

    def adam(self, p, g, m, v, lr, b1, b2, eps, step, grad_scale=1.0, hyper=None):
        """hyper: optional device tensor {lr/(1-b1^step), sqrt(1-b2^step)} read at run time (graph replay).  The launch is guarded
        by the persistent kernels' time-out word of this device: a step whose recurrent hand-off timed out (gradients invalid)
        updates nothing; `check_persistent()` then reports it."""
        self._check(self.lib.dlsg_adam(_p(p), _p(g), _p(m), _p(v), i64(p.numel()), f32(lr), f32(b1), f32(b2), f32(eps),
                                       int(step), f32(grad_scale), _p(hyper), _p(self._persist_word(p.device)), self._stream()), 'adam')
