import numbers


def old_div(a, b):
    """Python-2 division semantics, as `past.utils.old_div` defines them."""
    if isinstance(a, numbers.Integral) and isinstance(b, numbers.Integral):
        return a // b
    return a / b
