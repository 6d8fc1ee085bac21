from .ms_deform_attn_func import MSDeformAttnFunction, ms_deform_attn_backward, ms_deform_attn_forward  # noqa: F401
