"""llama.py:9 imports texttable (not installed here); only its dead reporting code would use it."""


class Texttable:
    def __init__(self, *a, **k):
        self.rows = []

    def header(self, row):
        self.rows.append(list(row))

    def add_row(self, row):
        self.rows.append(list(row))

    def draw(self):
        return "\n".join(" | ".join(str(c) for c in r) for r in self.rows)
