"""Alphabets used to size Less/Occ (reference: src/alphabets/mod.rs:49-133, dna.rs:23-35)."""


class Alphabet:
    def __init__(self, symbols):
        self.symbols = bytes(sorted(set(bytes(symbols))))

    def is_word(self, text):
        s = set(self.symbols)
        return all(c in s for c in bytes(text))

    def max_symbol(self):
        return self.symbols[-1] if self.symbols else None

    def insert(self, a):
        self.symbols = bytes(sorted(set(self.symbols) | {a}))

    def __len__(self):
        return len(self.symbols)

    def __bytes__(self):
        return self.symbols


class dna:
    @staticmethod
    def alphabet():
        return Alphabet(b"ACGTacgt")  # dna.rs:23-25

    @staticmethod
    def n_alphabet():
        return Alphabet(b"ACGTNacgtn")  # dna.rs:29-31

    @staticmethod
    def iupac_alphabet():
        return Alphabet(b"ACGTRYSWKMBDHVNacgtryswkmbdhvn")  # dna.rs:33-35
