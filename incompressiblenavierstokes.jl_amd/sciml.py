"""SciML-style right-hand side (sciml.jl:13-47): du/dt = project(bc(momentum(bc(u))))."""
from .operators import apply_bc_u_, momentum_
from .pressure import project_
from .setup import copyfield, scalarfield, vectorfield


def right_hand_side_(dudt, u, params, t):
    """In-place right-hand side (sciml.jl:35-47).  `params = (setup, psolver)`; `u` is not modified."""
    setup, psolver = params[0], params[1]
    p = scalarfield(setup)
    tmp = copyfield(u)
    apply_bc_u_(tmp, t, setup)
    momentum_(dudt, tmp, None, t, setup)
    apply_bc_u_(dudt, t, setup, dudt=True)
    project_(dudt, setup, psolver, p)
    return None


def create_right_hand_side(setup, psolver):
    """sciml.jl:13-19: returns `right_hand_side(u, param, t)` (allocating)."""

    def right_hand_side(u, param, t):
        dudt = vectorfield(setup)
        right_hand_side_(dudt, u, (setup, psolver), t)
        return dudt

    return right_hand_side
