"""host-side mirror of the reference's pyatac modules that sit on the occ + nuc path"""
