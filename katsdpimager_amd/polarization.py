"""Polarization bases and the Mueller matrices between them (host side of SURVEY 8f-1).

Same constants and results as ``katsdpimager.polarization`` (polarization.py:34-132): the
CASA enumeration of correlation products, each product's expansion in Stokes IQUV, and the
matrices that the preprocessing step (``preprocess.VisibilityCollectorDevice.add``) applies to
raw visibilities — either one matrix from the input products straight to the output Stokes
parameters, or a pair going through the circular frame when parallactic-angle rotation is
applied in between (frontend.py:675-680).
"""
import numpy as np

_PRODUCTS = ['I', 'Q', 'U', 'V', 'RR', 'RL', 'LR', 'LL', 'XX', 'XY', 'YX', 'YY']
#: CASA enumeration, 1-based (polarization.py:34-45); index 0 is unused
STOKES_NAMES = [None] + _PRODUCTS
(STOKES_I, STOKES_Q, STOKES_U, STOKES_V, STOKES_RR, STOKES_RL, STOKES_LR, STOKES_LL,
 STOKES_XX, STOKES_XY, STOKES_YX, STOKES_YY) = range(1, 13)
STOKES_IQUV = [STOKES_I, STOKES_Q, STOKES_U, STOKES_V]

# product = sum of coefficient * (I, Q, U, V)  (polarization.py:53-66)
_EXPANSION = {
    'I': (1, 0, 0, 0), 'Q': (0, 1, 0, 0), 'U': (0, 0, 1, 0), 'V': (0, 0, 0, 1),
    'RR': (1, 0, 0, 1), 'LL': (1, 0, 0, -1), 'RL': (0, 1, 1j, 0), 'LR': (0, 1, -1j, 0),
    'XX': (1, 1, 0, 0), 'YY': (1, -1, 0, 0), 'XY': (0, 0, 1, 1j), 'YX': (0, 0, 1, -1j),
}
STOKES_COEFF = np.array([(0, 0, 0, 0)] + [_EXPANSION[name] for name in _PRODUCTS], np.complex64)


def polarization_matrix(outputs, inputs):
    """Mueller matrix X, complex64 [len(outputs)][len(inputs)], with outputs = X @ inputs for
    every Stokes vector (polarization.py:69-105).  Raises ValueError when the inputs do not
    determine the outputs.  Entries within rounding of a multiple of 1/4 are snapped to it, so
    that structural zeros are exact zeros (the conversion kernel skips them)."""
    a = STOKES_COEFF[list(inputs), :]          # inputs  = a @ s
    b = STOKES_COEFF[list(outputs), :]         # outputs = b @ s
    # least-squares solution of x @ a = b
    x = np.linalg.lstsq(a.T, b.T, rcond=1e-5)[0].T
    if np.linalg.norm(x @ a - b) > 1e-5:
        raise ValueError('no solution')
    x = x.astype(np.complex64)
    snapped = (np.round(np.float32(4) * x) * np.float32(0.25)).astype(np.complex64)
    return np.where(np.isclose(x, snapped), snapped, x)


def polarization_matrices(outputs, inputs):
    """(circular -> outputs, inputs -> circular), polarization.py:108-132."""
    circular = [STOKES_RR, STOKES_RL, STOKES_LR, STOKES_LL]
    return polarization_matrix(outputs, circular), polarization_matrix(circular, inputs)
