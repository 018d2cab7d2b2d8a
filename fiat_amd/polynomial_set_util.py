def mis(m, n):
    """All m-tuples of non-negative integers summing to n, first entry descending
    (the ordering of derivative tables, FIAT/polynomial_set.py:23-32)."""
    if m == 1:
        return [(n,)]
    if n == 0:
        return [(0,) * m]
    return [(n - i,) + rest for i in range(n + 1) for rest in mis(m - 1, i)]
