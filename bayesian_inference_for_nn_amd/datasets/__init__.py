from .Dataset import ArrayDataset, Dataset
