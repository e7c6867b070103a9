"""Material presets (host).  Mirrors the reference's ``MatSet`` / ``Material``
(reference src/diffelastic/material_model.py:8-25): tuples (density, youngs, poisson, alpha, beta)."""


class MatSet:
    Ceramic = 2700, 7.2e10, 0.19, 6, 1e-7
    Glass = 2600, 6.2e10, 0.20, 1, 1e-7
    Wood = 750, 1.1e10, 0.25, 60, 2e-6
    Plastic = 1070, 1.4e9, 0.35, 30, 1e-6
    Iron = 8000, 2.1e11, 0.28, 10, 1e-7
    Polycarbonate = 1190, 2.4e9, 0.37, 0.5, 4e-7
    Steel = 7850, 2.0e11, 0.29, 20, 3e-8
    Tin = 7265, 5e10, 0.325, 2, 3e-8
    Test = 2700, 6e10, 0.19, 6, 1e-7
    RandomMin = 2700, 1e10, 0.1, 6, 1e-7
    RandomMax = 2700, 1e11, 0.4, 6, 1e-7


class Material(object):
    def __init__(self, material):
        self.density, self.youngs, self.poisson, self.alpha, self.beta = material
