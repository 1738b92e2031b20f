"""Experiment logger writing the files the reference's analysis code reads.

The reference logs through the third-party ``test_tube.Experiment`` (``fit`` calls
``exp.log(row)`` / ``exp.save()``, ``export_hparams`` calls ``exp.tag(hparams)``; reference
training.py:364-372,413, fitting/utils.py:756-777,866-871).  This class writes the same layout
without that dependency:

    <save_dir>/<name>/version_<K>/metrics.csv      one row per logged dict, union of all keys
    <save_dir>/<name>/version_<K>/meta_tags.csv    key,value rows of the tagged hparams

``version`` is the smallest unused K unless given.  ``get_best_model_version`` and the
reference's plotting read ``metrics.csv`` by column name (``val_loss``, ``tr_loss``, ``epoch``,
``dataset`` ...), which is all that is relied on here.
"""

import csv
import os
import time

__all__ = ['Experiment']


class Experiment(object):

    def __init__(self, name='default', debug=False, save_dir=None, version=None,
                 autosave=False, description=None):
        self.name = name
        self.debug = debug
        self.save_dir = save_dir if save_dir is not None else os.getcwd()
        self.autosave = autosave
        self.description = description
        self.metrics = []
        self.tags = {}
        self.created_at = time.time()
        root = os.path.join(self.save_dir, self.name)
        if version is None and not self.debug:
            # Claim the directory atomically: the ranks of a grid search (one grid point per
            # rank, usually the same experiment name) get here at the same time, and "list, then
            # create" would hand two of them the same version_K -- `mkdir` either creates the
            # directory or fails, so exactly one process owns each K.
            version = self._next_version(root)
            while True:
                try:
                    os.makedirs(self.get_data_path(self.name, version), exist_ok=False)
                    break
                except FileExistsError:
                    version += 1
        elif version is None:
            version = self._next_version(root)
        self.version = int(version)
        if not self.debug:
            os.makedirs(self.get_data_path(self.name, self.version), exist_ok=True)

    @staticmethod
    def _next_version(root):
        taken = []
        if os.path.isdir(root):
            for entry in os.listdir(root):
                if entry.startswith('version_') and entry[8:].isdigit():
                    taken.append(int(entry[8:]))
        return max(taken) + 1 if taken else 0

    def get_data_path(self, exp_name, exp_version):
        return os.path.join(self.save_dir, exp_name, 'version_%i' % exp_version)

    def log(self, metrics_dict):
        """Append one metric row (values are stored as given; ``created_at`` is added)."""
        if self.debug:
            return
        row = dict(metrics_dict)
        row.setdefault('created_at', time.strftime('%Y-%m-%d %H:%M:%S'))
        self.metrics.append(row)
        if self.autosave:
            self.save()

    def tag(self, tag_dict):
        if self.debug:
            return
        self.tags.update(tag_dict)
        if self.autosave:
            self.save()

    def save(self):
        """(Re)write ``metrics.csv`` and ``meta_tags.csv``."""
        if self.debug:
            return
        path = self.get_data_path(self.name, self.version)
        os.makedirs(path, exist_ok=True)
        columns = []
        for row in self.metrics:
            for key in row:
                if key not in columns:
                    columns.append(key)
        with open(os.path.join(path, 'metrics.csv'), 'w', newline='') as f:
            writer = csv.DictWriter(f, fieldnames=columns, restval='')
            writer.writeheader()
            for row in self.metrics:
                writer.writerow(row)
        with open(os.path.join(path, 'meta_tags.csv'), 'w', newline='') as f:
            writer = csv.writer(f)
            writer.writerow(['key', 'value'])
            for key, value in self.tags.items():
                writer.writerow([key, value])
