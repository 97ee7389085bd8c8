"""ctypes binding of libtdvc_hip.so (the C ABI declared in include/tdvc.h).

The library is loaded on first use and the load FAILS LOUDLY when the shared object is
missing: there is no CPU / ATen fallback behind these operators. Build it with
`python -c "import __graft_entry__ as g; g.build()"` or `make -C td-vc-gan_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('TDVC_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libtdvc_hip.so')   # override: A/B builds in tools/

c_float_p = C.POINTER(C.c_float)


class Xform(C.Structure):
    _fields_ = [('kind', C.c_int32), ('slope', C.c_float), ('scale', C.c_float),
                ('aux', C.c_void_p), ('aux_bs', C.c_int64)]


class ConvDesc(C.Structure):
    _fields_ = [('kind', C.c_int32), ('B', C.c_int32), ('Cin', C.c_int32), ('Cout', C.c_int32),
                ('Tin', C.c_int32), ('Tout', C.c_int32), ('K', C.c_int32), ('stride', C.c_int32),
                ('dilation', C.c_int32), ('pad', C.c_int32), ('groups', C.c_int32), ('reflect', C.c_int32),
                ('w_cin', C.c_int32), ('w_cin_off', C.c_int32)]


class ConvFwdArgs(C.Structure):
    _fields_ = [('x', C.c_void_p), ('x_bs', C.c_int64), ('x_xf', Xform), ('w', C.c_void_p), ('bias', C.c_void_p),
                ('res', C.c_void_p), ('res_bs', C.c_int64), ('post_act', C.c_int32), ('post_slope', C.c_float),
                ('out_scale', C.c_float), ('add', C.c_void_p), ('add_bs', C.c_int64), ('y', C.c_void_p),
                ('y_bs', C.c_int64), ('bias3', C.c_void_p), ('sign_bits', C.c_void_p), ('sign_bits_bs', C.c_int64)]


class FilmBlockArgs(C.Structure):
    _fields_ = [('B', C.c_int32), ('C', C.c_int32), ('T', C.c_int32), ('K', C.c_int32), ('dilation', C.c_int32),
                ('x', C.c_void_p), ('x_bs', C.c_int64), ('w1', C.c_void_p), ('b1', C.c_void_p), ('h', C.c_void_p), ('h_bs', C.c_int64),
                ('gb', C.c_void_p), ('gb_bs', C.c_int64), ('w2', C.c_void_p), ('b2', C.c_void_p), ('add', C.c_void_p), ('add_bs', C.c_int64),
                ('scale', C.c_float), ('slope', C.c_float), ('y', C.c_void_p), ('y_bs', C.c_int64)]


class FilmCondArgs(C.Structure):
    _fields_ = [('B', C.c_int32), ('T', C.c_int32), ('n_cond', C.c_int32), ('n_var', C.c_int32), ('C2', C.c_int32),
                ('exc', C.c_void_p), ('exc_bs', C.c_int64), ('w0', C.c_void_p), ('k3', C.c_void_p),
                ('w2', C.c_void_p), ('b2', C.c_void_p), ('cv0', C.c_void_p), ('cv0_bs', C.c_int64),
                ('gb', C.c_void_p), ('gb_bs', C.c_int64), ('slope', C.c_float)]


class FilmCond0BwdArgs(C.Structure):
    _fields_ = [('B', C.c_int32), ('T', C.c_int32), ('n_cond', C.c_int32), ('n_var', C.c_int32),
                ('dcv', C.c_void_p), ('dcv_bs', C.c_int64), ('exc', C.c_void_p), ('exc_bs', C.c_int64), ('w0', C.c_void_p),
                ('dexc', C.c_void_p), ('dexc_bs', C.c_int64), ('dk3', C.c_void_p), ('dw0', C.c_void_p),
                ('workspace', C.c_void_p), ('workspace_bytes', C.c_size_t)]


class FilmCondBwdArgs(C.Structure):
    _fields_ = [('B', C.c_int32), ('T', C.c_int32), ('n_cond', C.c_int32), ('n_var', C.c_int32), ('C2', C.c_int32),
                ('dgb', C.c_void_p), ('dgb_bs', C.c_int64), ('wt2', C.c_void_p),
                ('cv0_sign_bits', C.c_void_p), ('cv0_sign_bits_bs', C.c_int64), ('cv0', C.c_void_p), ('cv0_bs', C.c_int64),
                ('exc', C.c_void_p), ('exc_bs', C.c_int64), ('w0', C.c_void_p),
                ('dexc', C.c_void_p), ('dexc_bs', C.c_int64), ('dk3', C.c_void_p), ('dw0', C.c_void_p),
                ('workspace', C.c_void_p), ('workspace_bytes', C.c_size_t), ('slope', C.c_float)]


class ConvDgradArgs(C.Structure):
    _fields_ = [('dy', C.c_void_p), ('dy_bs', C.c_int64), ('dy_xf', Xform), ('w', C.c_void_p), ('wt', C.c_void_p),
                ('epilogue', C.c_int32), ('x_in', C.c_void_p), ('x_in_bs', C.c_int64), ('slope', C.c_float),
                ('gb', C.c_void_p), ('gb_bs', C.c_int64), ('dgb', C.c_void_p), ('dgb_bs', C.c_int64),
                ('add', C.c_void_p), ('add_bs', C.c_int64), ('add_scale', C.c_float),
                ('dx', C.c_void_p), ('dx_bs', C.c_int64), ('x_sign_bits', C.c_void_p), ('x_sign_bits_bs', C.c_int64)]


class L1Pair(C.Structure):
    _fields_ = [('a', C.c_void_p), ('b', C.c_void_p), ('da', C.c_void_p), ('n', C.c_int64), ('weight', C.c_float)]


class ConvWgradArgs(C.Structure):
    _fields_ = [('x', C.c_void_p), ('x_bs', C.c_int64), ('x_xf', Xform), ('dy', C.c_void_p), ('dy_bs', C.c_int64),
                ('dy_xf', Xform), ('dw', C.c_void_p), ('dbias', C.c_void_p), ('workspace', C.c_void_p),
                ('workspace_bytes', C.c_size_t)]


EUNSUPPORTED = -4      # tdvc_status TDVC_EUNSUPPORTED
XF_NONE, XF_LRELU, XF_FILM_LRELU, XF_MASK_LRELU, XF_MASK_TANH = range(5)
CONV, CONV_TRANSPOSE = 0, 1
POST_NONE, POST_LRELU, POST_TANH = 0, 1, 2
DG_PLAIN, DG_MASK_LRELU, DG_FILM = 0, 1, 2

# name -> (restype, argtypes); every symbol include/tdvc.h declares
_vp, _i, _f, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_int64
SIGNATURES = {
    'tdvc_conv_fwd': (_i, [C.POINTER(ConvDesc), C.POINTER(ConvFwdArgs), _vp]),
    'tdvc_conv_dgrad': (_i, [C.POINTER(ConvDesc), C.POINTER(ConvDgradArgs), _vp]),
    'tdvc_conv_wgrad': (_i, [C.POINTER(ConvDesc), C.POINTER(ConvWgradArgs), _vp]),
    'tdvc_conv_wgrad_workspace': (C.c_size_t, [C.POINTER(ConvDesc)]),
    'tdvc_conv_x6_weight_planes_bytes': (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    'tdvc_conv_x6_weight_planes': (_i, [_vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp]),
    'tdvc_conv_fwd_x6': (_i, [C.POINTER(ConvDesc), C.POINTER(ConvFwdArgs), _vp, _vp]),
    'tdvc_film_cond_fwd': (_i, [C.POINTER(FilmCondArgs), _vp]),
    'tdvc_film_cond_fwd_x6': (_i, [C.POINTER(FilmCondArgs), C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    'tdvc_film_cond0_bwd': (_i, [C.POINTER(FilmCond0BwdArgs), _vp]),
    'tdvc_film_cond_bwd': (_i, [C.POINTER(FilmCondBwdArgs), _vp]),
    'tdvc_film_cond_bwd_workspace': (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'tdvc_film_block_fwd': (_i, [_vp, _vp]),
    'tdvc_film_k3_fwd': (_i, [_vp, _i64, _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp]),
    'tdvc_film_k3_bwd': (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp]),
    'tdvc_film_k3_multi_fwd': (_i, [_vp, _i64, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'tdvc_film_k3_multi_bwd': (_i, [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'tdvc_film_cond0_bwd_workspace': (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'tdvc_set_force_generic': (None, [_i]),
    'tdvc_debug_force_tile': (None, [_i]),
    'tdvc_debug_lds_cap': (None, [_i]),
    'tdvc_debug_knob': (None, [_i, _i]),
    'tdvc_fold_defer': (None, [_i]),
    'tdvc_fold_flush': (_i, [C.c_void_p]),
    'tdvc_fold_reset': (None, [C.c_void_p]),
    'tdvc_debug_poison_lds': (_i, [C.c_uint32, C.c_void_p]),
    'tdvc_debug_marker': (_i, [C.c_void_p]),
    'tdvc_debug_trace': (None, [_i]),
    'tdvc_debug_trace_dump': (C.c_size_t, [C.c_char_p, C.c_size_t]),
    'tdvc_weight_norm_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    'tdvc_weight_norm_fwd_t': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    'tdvc_weight_norm_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    'tdvc_adamw': (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _i, _vp, _f, _vp]),
    'tdvc_adamw_clipped': (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _i, _vp, _f, _vp, _vp]),
    'tdvc_grad_clip_coef': (_i, [_vp, _i64, _f, _f, _vp, _vp, _vp]),
    'tdvc_roll_batches': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    'tdvc_inc_i32': (_i, [_vp, C.c_int32, _vp]),
    'tdvc_l2norm_fwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    'tdvc_l2norm_bwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'tdvc_gather_ch_fwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    'tdvc_gather_ch_bwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    'tdvc_concat_cond': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'tdvc_concat_cond_bwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'tdvc_edge_sum3': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'tdvc_axpby': (_i, [_vp, _vp, _vp, _f, _f, _i64, _vp]),
    'tdvc_fill': (_i, [_vp, _f, _i64, _vp]),
    'tdvc_sum_n': (_i, [_vp, C.c_int32, _vp, _i64, _vp]),
    'tdvc_gate_fwd': (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i, _i, _i, _vp]),
    'tdvc_gate_bwd': (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i, _i, _i, _vp]),
    'tdvc_cin_fwd': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    'tdvc_cin_bwd': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'tdvc_mse_const_fwd': (_i, [_vp, _i64, _f, _f, _vp, _vp]),
    'tdvc_mse_const_bwd': (_i, [_vp, _i64, _f, _f, _vp, _vp, _vp]),
    'tdvc_l1_fwd': (_i, [_vp, _vp, _i64, _f, _vp, _vp]),
    'tdvc_l1_bwd': (_i, [_vp, _vp, _i64, _f, _vp, _vp, _i, _vp]),
    'tdvc_l1_multi_fwd': (_i, [_vp, _i, _vp, _vp]),
    'tdvc_l1_multi_bwd': (_i, [_vp, _i, _vp, _vp]),
    'tdvc_reflect_pad_fwd': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'tdvc_reflect_pad_bwd': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'tdvc_power_fwd': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'tdvc_power_bwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    'tdvc_log_l1_fwd': (_i, [_vp, _vp, _i64, _f, _f, _vp, _vp]),
    'tdvc_log_l1_bwd': (_i, [_vp, _vp, _i64, _f, _f, _vp, _vp, _vp]),
    'tdvc_cross_entropy_fwd': (_i, [_vp, _vp, _i, _i, _f, _vp, _vp, _vp]),
    'tdvc_cross_entropy_bwd': (_i, [_vp, _vp, _i, _i, _f, _vp, _vp, _vp]),
    'tdvc_f0_to_excitation': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _vp]),
    'tdvc_contrastive_fwd_bwd': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _vp]),
    'tdvc_last_error': (C.c_char_p, []),
    'tdvc_version': (_i, []),
}

_lib = None


class TdvcError(RuntimeError):
    pass


def lib():
    """The loaded library; raises if libtdvc_hip.so has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TdvcError(f'{LIB_PATH} is missing: build the HIP extension first '
                            f'(python -c "import __graft_entry__ as g; g.build()"). '
                            'There is no CPU fallback for this path.')
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)      # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = l
        for kv in filter(None, os.environ.get('TDVC_KNOBS', '').split(',')):      # diagnostic: "5=1,6=1" presets tdvc_debug_knob values
            k, v = kv.split('=')
            l.tdvc_debug_knob(int(k), int(v))
    return _lib


def check(rc):
    if rc != 0:
        raise TdvcError(f'tdvc call failed ({rc}): {lib().tdvc_last_error().decode()}')


def traced_kernels():
    """Names recorded since tdvc_debug_trace(1), normalised like rocprofv3's kernel names without spaces:
    'conv_lean_kernel<1,4,1,4,0,0>'."""
    l = lib()
    n = l.tdvc_debug_trace_dump(None, 0)
    buf = C.create_string_buffer(n)
    l.tdvc_debug_trace_dump(buf, n)
    return {normalize_kernel_name(s) for s in buf.value.decode().split('\n') if s}


def normalize_kernel_name(s):
    """'void tdvc::conv_lean_kernel<3, 4, 1, 4, 0, 1>(tdvc::LeanP)' -> 'conv_lean_kernel<3,4,1,4,0,1>'."""
    s = s.strip().strip('"').replace('(anonymous namespace)::', '')
    if s.startswith('void '):
        s = s[5:]
    depth, cut = 0, len(s)
    for i, ch in enumerate(s):      # drop the argument list: first '(' outside template brackets
        if ch == '<':
            depth += 1
        elif ch == '>':
            depth -= 1
        elif ch == '(' and depth == 0:
            cut = i
            break
    s = s[:cut].replace(' ', '').replace('tdvc::', '').replace('(anonymousnamespace)::', '')
    if s.startswith('conv_lean_kernel<') and s.endswith(',false>'):      # the defaulted FOLD = false argument is not part of the instance's name
        s = s[:-len(',false>')] + '>'
    return s
