"""host-side mirror of the reference's nucleoatac modules that sit on the occ + nuc path"""
