"""VTK output of P1-DG block functions (reference: ``DuneGDTVisualizer`` at discretize_elliptic_block_swipdg.py:802 and
``grid.visualize`` at grid.py:35; SURVEY.md section 8f "next" #4).

Legacy ASCII ``.vtk`` unstructured grids, readable by ParaView: every triangle carries its own three points, so the
discontinuous function is represented exactly (POINT_DATA), the subdomain index is CELL_DATA.  Host-side I/O only."""
import numpy as np


def _mesh(grid, subdomains):
    t = grid.template
    org = np.stack([grid.subdomain_origin(int(ii)) for ii in subdomains])            # [S, 2]
    pts = (org[:, None, None, :] + t.points[None]).reshape(-1, 2)                    # [S * n_T * 3, 2]
    ncell = len(subdomains) * t.n_T
    return pts, ncell


def write_vtk(filename, grid, subdomains, point_data=None, title='pylrbms_amd'):
    """``point_data``: dict name -> array [len(subdomains), n] (one value per local DoF).  Returns the file name."""
    subdomains = [int(ii) for ii in subdomains]
    pts, ncell = _mesh(grid, subdomains)
    t = grid.template
    if not filename.endswith('.vtk'):
        filename += '.vtk'
    with open(filename, 'w') as f:
        f.write('# vtk DataFile Version 3.0\n{}\nASCII\nDATASET UNSTRUCTURED_GRID\n'.format(title))
        f.write('POINTS {} double\n'.format(len(pts)))
        np.savetxt(f, np.column_stack([pts, np.zeros(len(pts))]), fmt='%.17g')
        f.write('CELLS {} {}\n'.format(ncell, 4 * ncell))
        conn = np.arange(3 * ncell).reshape(ncell, 3)
        np.savetxt(f, np.column_stack([np.full(ncell, 3), conn]), fmt='%d')
        f.write('CELL_TYPES {}\n'.format(ncell))
        np.savetxt(f, np.full(ncell, 5), fmt='%d')                                   # VTK_TRIANGLE
        f.write('CELL_DATA {}\nSCALARS subdomain int 1\nLOOKUP_TABLE default\n'.format(ncell))
        np.savetxt(f, np.repeat(np.asarray(subdomains), t.n_T), fmt='%d')
        if point_data:
            f.write('POINT_DATA {}\n'.format(len(pts)))
            for name, vals in point_data.items():
                vals = np.asarray(vals, dtype=np.float64).reshape(-1)
                assert len(vals) == len(pts), (name, vals.shape, len(pts))
                f.write('SCALARS {} double 1\nLOOKUP_TABLE default\n'.format(name))
                np.savetxt(f, vals, fmt='%.17g')
    return filename


def visualize_block_array(U, grid, subdomains, filename, name='u'):
    """One file per vector of the block array ``U`` (``filename.vtk`` for a single vector, else ``filename_k.vtk``)."""
    data = U.tensor.cpu().numpy()                                                   # [S, n, len]
    out = []
    for k in range(data.shape[2]):
        fn = filename if data.shape[2] == 1 else '{}_{}'.format(filename, k)
        out.append(write_vtk(fn, grid, subdomains, {name: data[:, :, k]}))
    return out
