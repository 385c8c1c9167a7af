def cmp(x, y):
    return (x > y) - (x < y)
