"""Minimal stand-in for the (uninstalled) `primefac` package, used ONLY by
tests/golden/make_golden.py when it imports the reference from /root/reference.
The reference calls primefac.primefac(n) in method.py:17 and needs the prime
factors in ascending order."""


def primefac(n):
    n = int(n)
    f = 2
    while f * f <= n:
        while n % f == 0:
            yield f
            n //= f
        f += 1 if f == 2 else 2
    if n > 1:
        yield n
