"""bfcnn/regularizers.py as the reference's tests/bfcnn/test_regularizer.py uses it -- `reshape_to_2d`, `wt_x_w`, the keras-style
regulariser objects (`L1`, `L2`, `L1L2`, `SoftOrthonormalConstraintRegularizer`, `SoftOrthogonalConstraintRegularizer`,
`RegularizerMixer`) and `builder` -- on device tensors over the C-ABI operators the training steps use
(`bf_op_reg_elementwise`, `bf_op_reg_soft_orthogonal_ex`, `bf_op_transpose2d`, `bf_op_matmul_wgrad`).  A regulariser called on a
kernel returns its value as a 0-d device tensor; inside `train_step_single_gpu` the same operators also add the gradients."""
from enum import Enum
from typing import Dict, List, Union

import torch

from . import _native as N
from .constants import CONFIG_STR, TYPE_STR

REGULARIZERS_STR = "regularizers"


class RegularizationType(Enum):
    """regularizers.py:19-44"""
    L1 = 0
    L2 = 1
    L1L2 = 2
    SOFT_ORTHONORMAL = 3
    SOFT_ORTHOGONAL = 4

    @staticmethod
    def from_string(type_str: str) -> "RegularizationType":
        if type_str is None:
            raise ValueError("type_str must not be null")
        if not isinstance(type_str, str):
            raise ValueError("type_str must be string")
        type_str = type_str.strip().upper()
        if len(type_str) <= 0:
            raise ValueError("stripped type_str must not be empty")
        return RegularizationType[type_str]

    def to_string(self) -> str:
        return self.name


def _gpu(x) -> torch.Tensor:
    if not isinstance(x, torch.Tensor) or not x.is_cuda:
        raise RuntimeError("regularisers run on MI355X device tensors: there is no CPU execution path")
    return x.to(torch.float32).contiguous()


def reshape_to_2d(weights: torch.Tensor) -> torch.Tensor:
    """regularizers.py:159-190: [a, b] -> [b, a]; an HWIO kernel [kh, kw, cin, cout] -> [cout, kh kw cin]; other ranks unchanged"""
    w = _gpu(weights)
    if w.dim() not in (2, 4):
        return w
    cout = int(w.shape[-1])
    rows = w.numel() // cout
    out = torch.empty((cout, rows), dtype=torch.float32, device=w.device)
    N.check(N.lib().bf_op_transpose2d(N.ptr(w), N.ptr(out), rows, cout, N.stream_ptr(w)), None, "bf_op_transpose2d")
    return out


def wt_x_w(weights: torch.Tensor) -> torch.Tensor:
    """regularizers.py:196-206: Wt Wt^T with Wt = reshape_to_2d(weights): the [cout, cout] Gram matrix of the output channels"""
    w = _gpu(weights)
    if w.dim() not in (2, 4):
        raise ValueError("wt_x_w takes a rank-2 or rank-4 kernel")
    cout = int(w.shape[-1])
    rows = w.numel() // cout
    out = torch.empty((cout, cout), dtype=torch.float32, device=w.device)
    scratch = torch.empty(max(64 * cout * cout, 1024), dtype=torch.float32, device=w.device)
    N.check(N.lib().bf_op_matmul_wgrad(N.ptr(w), N.ptr(w), N.ptr(out), rows, cout, cout, N.ptr(scratch), scratch.numel(), N.stream_ptr(w)),
            None, "bf_op_matmul_wgrad")
    return out


class Regularizer:
    def __call__(self, x) -> torch.Tensor:
        raise NotImplementedError

    def get_config(self) -> Dict:
        return {}


class L1L2(Regularizer):
    """keras.regularizers.L1L2: l1 sum |x| + l2 sum x^2"""

    def __init__(self, l1: float = 0.0, l2: float = 0.0):
        self.l1, self.l2 = float(l1), float(l2)

    def __call__(self, x):
        w = _gpu(x)
        value = torch.zeros(1, dtype=torch.float32, device=w.device)
        for kind, coef in ((N.BF_REG_L1, self.l1), (N.BF_REG_L2, self.l2)):
            if coef:
                N.check(N.lib().bf_op_reg_elementwise(N.ptr(w), None, w.numel(), kind, coef, 0.0, N.ptr(value), N.stream_ptr(w)), None,
                        "bf_op_reg_elementwise")
        return value[0]

    def get_config(self):
        return {"l1": self.l1, "l2": self.l2}


class L1(L1L2):
    def __init__(self, l1: float = 0.01):
        super().__init__(l1=l1, l2=0.0)


class L2(L1L2):
    def __init__(self, l2: float = 0.01):
        super().__init__(l1=0.0, l2=l2)


class _SoftGram(Regularizer):
    _mask_diagonal = 0

    def __init__(self, lambda_coefficient: float, l1_coefficient: float, l2_coefficient: float, **kwargs):
        self._lambda_coefficient, self._l1_coefficient, self._l2_coefficient = float(lambda_coefficient), float(l1_coefficient), float(l2_coefficient)

    def __call__(self, x):
        w = _gpu(x)
        if w.dim() not in (2, 4):
            raise ValueError("a rank-2 or rank-4 kernel is expected")
        cout = int(w.shape[-1])
        rows = w.numel() // cout
        value = torch.zeros(1, dtype=torch.float32, device=w.device)
        scratch = torch.empty(2 * cout * cout + 64, dtype=torch.float32, device=w.device)
        N.check(N.lib().bf_op_reg_soft_orthogonal_ex(N.ptr(w), None, rows, cout, max(self._lambda_coefficient, 0.0), max(self._l1_coefficient, 0.0),
                                                     max(self._l2_coefficient, 0.0), 0.0, N.ptr(value), N.ptr(scratch), self._mask_diagonal,
                                                     N.stream_ptr(w)), None, "bf_op_reg_soft_orthogonal_ex")
        return value[0]

    def get_config(self):
        return {"lambda_coefficient": self._lambda_coefficient, "l1_coefficient": self._l1_coefficient, "l2_coefficient": self._l2_coefficient}


class SoftOrthogonalConstraintRegularizer(_SoftGram):
    """regularizers.py:208-280: lambda ||G o (1 - I)||_F^2 + l1 sum |G o (1 - I)| + l2 sum (G o (1 - I))^2, G = wt_x_w(x)"""
    _mask_diagonal = 1

    def __init__(self, lambda_coefficient: float = 1.0, l1_coefficient: float = 0.01, l2_coefficient: float = 0.00, **kwargs):
        super().__init__(lambda_coefficient, l1_coefficient, l2_coefficient)


class SoftOrthonormalConstraintRegularizer(_SoftGram):
    """regularizers.py:283-338: lambda ||G - I||_F^2 + l1 sum |G| + l2 sum G^2"""
    _mask_diagonal = 0

    def __init__(self, lambda_coefficient: float = 1.0, l1_coefficient: float = 0.001, l2_coefficient: float = 0.00, **kwargs):
        super().__init__(lambda_coefficient, l1_coefficient, l2_coefficient)


class RegularizerMixer(Regularizer):
    """regularizers.py:49-76: the sum of its regularisers"""

    def __init__(self, regularizers: List[Regularizer]):
        self._regularizers = regularizers

    def __call__(self, x):
        total = None
        for r in self._regularizers:
            v = r(x)
            if total is None:
                total = v.clone()
            else:
                N.check(N.lib().bf_op_axpy(N.ptr(total), N.ptr(v), 1.0, 0, 1, N.stream_ptr(total)), None, "bf_op_axpy")
        return total

    def get_config(self):
        return {REGULARIZERS_STR: [r.get_config() for r in self._regularizers]}


def builder_helper(config: Union[str, Dict, Regularizer], verbose: bool = False) -> Regularizer:
    """regularizers.py:81-130"""
    if config is None:
        raise ValueError("config cannot be None")
    if isinstance(config, str):
        regularizer_type, params = config.lower(), {}
    elif isinstance(config, dict):
        regularizer_type, params = config.get(TYPE_STR, None).lower(), config.get(CONFIG_STR, {})
    elif isinstance(config, Regularizer) and type(config) is not Regularizer:
        return config
    else:
        raise ValueError("don't know how to handle config")
    kind = RegularizationType.from_string(regularizer_type)
    return {RegularizationType.L1: L1, RegularizationType.L2: L2, RegularizationType.L1L2: L1L2,
            RegularizationType.SOFT_ORTHONORMAL: SoftOrthonormalConstraintRegularizer,
            RegularizationType.SOFT_ORTHOGONAL: SoftOrthogonalConstraintRegularizer}[kind](**params)


def builder(config: Union[str, Dict, List]) -> Regularizer:
    """regularizers.py:133-154: one regulariser, or a RegularizerMixer of a list"""
    if config is None:
        raise ValueError("config cannot be None")
    if isinstance(config, list):
        return RegularizerMixer(regularizers=[builder_helper(config=r) for r in config])
    return builder_helper(config=config)
